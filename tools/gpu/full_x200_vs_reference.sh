#!/bin/bash
# The whole genome x 200 betas (north_star's named bit-exactness target) against the reference binary, on the library in the tree:
#   bash tools/gpu/full_x200_vs_reference.sh [samples]      -> gpurun_out/full_vs_reference_x<samples>.log
set -u
N=${1:-200}
mkdir -p gpurun_out
timeout 1500 python tools/full_vs_reference.py --samples $N > gpurun_out/full_vs_reference_x$N.log 2>&1
echo "full x$N vs reference: rc $? ($(tail -1 gpurun_out/full_vs_reference_x$N.log | cut -c1-300))"
