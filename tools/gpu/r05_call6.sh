#!/bin/bash
# round 5, call 6: (a) x200 end to end: where do 0.83 s of "segmentation" go when the upload alone should take 0.34 s?  Upload threads / piece sweep with WGBSSEG_PROFILE=1;
# (b) k_dp with the rows dealt unevenly (the worker on the recurrence wavefront's SIMD is the late one: fewer rows for it)
set -u
O=$PWD/gpurun_out/r05c6; mkdir -p $O
for u in 2 4 6; do for nch in 480 240; do
  echo "#### uneven$u chunks $nch" >> $O/dp_uneven.txt
  timeout 200 python tools/dp_timing.py $nch 8 --flags "-DWGBSSEG_DP_UNEVEN=$u" --tag uneven$u 2>&1 | grep -v "WGBSSEG_LIB\|amdgpu.ids" | tail -9 | cut -c1-300 >> $O/dp_uneven.txt
done; done
grep "####\|recurrence wavefront\|barrier wait\|load-issue" $O/dp_uneven.txt | cut -c1-230
timeout 900 python tools/e2e_bench.py --samples 200 --keep > $O/e2e_x200_default.log 2>&1; echo "x200 default: rc $?"; grep "^run\|BED\|inputs" $O/e2e_x200_default.log | cut -c1-420
sync
for spec in "synced:WGBSSEG_NOP=1" "t8:WGBSSEG_UPLOAD_THREADS=8" "t16:WGBSSEG_UPLOAD_THREADS=16" "t8_4MB:WGBSSEG_UPLOAD_THREADS=8 WGBSSEG_UPLOAD_PIECE_KB=4096" "t2:WGBSSEG_UPLOAD_THREADS=2"; do
  n=${spec%%:*}; e=${spec#*:}
  env $e timeout 600 python tools/e2e_bench.py --samples 200 --keep > $O/e2e_x200_$n.log 2>&1; echo "$n: rc $?"; grep "^run\|BED" $O/e2e_x200_$n.log | cut -c1-420
done
rm -rf /tmp/wgbs_e2e
