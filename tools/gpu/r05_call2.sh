#!/bin/bash
# round 5, call 2: what costs the recurrence wavefront its 14 ticks per step in situ (33 in the bare chain, 47 in the kernel)?  Timing build with the workers
# switched off piece by piece (WGBSSEG_DP_DEBUG: 1 workers leave, 2 no LDS commits, 4 no row loads; results are wrong in these modes, only the clock counts);
# then the rewritten pat counting kernel against the reference binary.
set -u
REPO=$PWD
O=$REPO/gpurun_out/r05c2; mkdir -p $O
for nch in 240 480; do for m in 0 1 2 4 6; do
  echo "#### chunks $nch WGBSSEG_DP_DEBUG=$m" >> $O/dp_debug.txt
  WGBSSEG_DP_DEBUG=$m timeout 200 python tools/dp_timing.py $nch 8 2>&1 | grep -v "WGBSSEG_LIB\|amdgpu.ids" | tail -7 | cut -c1-330 >> $O/dp_debug.txt
done; done
grep "####\|recurrence wavefront" $O/dp_debug.txt | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_pat2beta.py -q -x > $O/tests_pat.log 2>&1; echo "pat tests: rc $? ($(tail -1 $O/tests_pat.log))"
