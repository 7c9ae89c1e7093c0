#!/bin/bash
set -u
O=gpurun_out/r04c4; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x > $O/parity.log 2>&1; echo "parity rc $? $(tail -1 $O/parity.log)"
bash tools/gpu/ab.sh r04c4 "ilp1 main main@WGBSSEG_TI=128 main@WGBSSEG_TI=128,WGBSSEG_CMAP128=1 main@WGBSSEG_TI=64" "--samples 8;--samples 16;--samples 32;--sites 3527181"
