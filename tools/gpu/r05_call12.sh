#!/bin/bash
# round 5, call 12: k_dp_lin (the lean recurrence kernel with segment loads + scatter commit, -inf put back by the recurrence wavefront): parity, timing build, bench A/B
set -u
O=$PWD/gpurun_out/r05c12; mkdir -p $O
WGBSSEG_DP_LIN=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "test_05 or test_06 or test_08 or test_09 or test_17 or test_07" > $O/tests_lin.log 2>&1; echo "parity (k_dp_lin): rc $? ($(tail -1 $O/tests_lin.log))"
tail -15 $O/tests_lin.log | cut -c1-200
for nch in 480 240; do
  WGBSSEG_DP_LIN=1 timeout 200 python tools/dp_timing.py $nch 8 2>&1 | grep -v "WGBSSEG_LIB\|amdgpu.ids" | tail -9 | cut -c1-300 > $O/dp_lin_$nch.txt; grep "recurrence wavefront\|barrier wait\|alone\|shared" $O/dp_lin_$nch.txt | tail -5 | cut -c1-230
done
bash tools/gpu/ab.sh r05c12 "main main@WGBSSEG_DP_LIN=1" "--samples 8;--samples 32;--sites 3527181" 2>&1 | tee $O/dp_lin_ab.txt
