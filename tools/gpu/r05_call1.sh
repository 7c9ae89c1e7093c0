#!/bin/bash
# round 5, first GPU call: SIMD-sharing micro-benchmark of the recurrence chain, in-situ placement / timing probes of k_dp (product order and the
# CU-parity selection of the recurrence wavefront), the new GPU tests, a baseline bench line of this box, the group-of-8 line, the whole genome x200
# against the reference binary.  Everything lands in gpurun_out/r05c1/.
set -u
REPO=$PWD
O=$REPO/gpurun_out/r05c1; mkdir -p $O
timeout 120 tools/micro/_build/dp_simd_share > $O/dp_simd_share.txt 2>&1; echo "micro: rc $?"
timeout 300 python tools/dp_timing.py 480 8 > $O/dp_timing_480.txt 2>&1; echo "dp_timing 480: rc $?"
timeout 300 python tools/dp_timing.py 240 8 > $O/dp_timing_240.txt 2>&1
timeout 300 python tools/dp_timing.py 483 32 > $O/dp_timing_483x32.txt 2>&1
tail -12 $O/dp_timing_480.txt $O/dp_timing_recsel_480.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "test_19 or test_11 or test_06" > $O/tests_new.log 2>&1; echo "new tests: rc $? ($(tail -1 $O/tests_new.log))"
timeout 600 python bench.py --steps 20 --warmup 5 2> $O/bench.err | tail -1 > $O/bench_base.json; echo "bench: $(cut -c1-200 $O/bench_base.json)"; python tools/summ.py $O/bench_base.json
timeout 600 python bench.py --gpus 8 --steps 5 --warmup 1 2> $O/group8.err | tail -1 > $O/group8_on_one_gpu.json; echo "group8: $(cut -c1-300 $O/group8_on_one_gpu.json)"
bash tools/gpu/full_x200_vs_reference.sh 200; cp gpurun_out/full_vs_reference_x200.log $O/
