#!/bin/bash
# What to run first when a GPU is at hand (from the repository root, e.g. through gpurun), in the order of what it protects:
#   1. the whole GPU suite                          (~5 min)
#   2. the boundary-aligned and the uniform fuzz     (SECONDS each, default 90: parity beyond the suite, fresh seeds by the clock)
#   3. the default bench line                        (~1 min with its CPU baseline)
# Everything lands in gpurun_out/checklist/.  Usage: tools/gpu_checklist.sh [SECONDS_PER_FUZZ]
set -u
S=${1:-90}
O=gpurun_out/checklist
mkdir -p $O
first=$(( ($(date +%s) % 100000) * 10 ))
timeout 900 python -m pytest tests -q -x -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests: rc $? ($(tail -1 $O/gpu_tests.log))"
timeout $((S + 30)) python tools/aligned_fuzz.py $first 1000000 $S 2>&1 | grep -v amdgpu.ids > $O/aligned_fuzz.log; echo "aligned fuzz: $(tail -1 $O/aligned_fuzz.log)"
timeout $((S + 30)) python tools/extra_fuzz.py $first 1000000 $S 2>&1 | grep -v amdgpu.ids > $O/extra_fuzz.log; echo "uniform fuzz: $(tail -1 $O/extra_fuzz.log)"
timeout 600 python bench.py 2> $O/bench.err | tail -1 > $O/bench.json; echo "bench: $(cut -c1-200 $O/bench.json)"
