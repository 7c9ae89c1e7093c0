#!/bin/bash
# round 3, GPU call 2: medium tiles + windows beyond 8000 sites: parity, then the islands workload with the medium class on / off and the carry stores with / without the non-temporal hint
set -u
O=gpurun_out/c2
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "test_16 or test_15 or test_11 or test_05 or test_06 or test_07 or test_08 or test_09 or test_13" > $O/parity_subset.log 2>&1; echo "parity subset: rc $? ($(tail -1 $O/parity_subset.log))"
WGBSSEG_FUZZ_SECONDS=40 timeout 300 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -s -k "time_boxed or reference_binary" > $O/fuzz.log 2>&1; echo "fuzz: rc $? ($(tail -1 $O/fuzz.log)) $(grep 'aligned fuzz' $O/fuzz.log)"
B="--matrix 0 --cpu-seconds 0 --e2e 0 --block-sums 0 --steps 8 --warmup 2"
timeout 300 python bench.py --islands $B 2> $O/isl_medium.err | tail -1 > $O/isl_medium.json
WGBSSEG_MEDIUM_WMAX=0 timeout 300 python bench.py --islands $B 2> /dev/null | tail -1 > $O/isl_nomedium.json
WGBSSEG_MEDIUM_WMAX=124 timeout 300 python bench.py --islands $B 2> /dev/null | tail -1 > $O/isl_medium124.json
WGBSSEG_NSM=16 timeout 300 python bench.py --islands $B 2> /dev/null | tail -1 > $O/isl_medium_ns16.json
WGBSSEG_MEDIUM_WMAX=0 WGBSSEG_SCAN_NT=1 timeout 300 python bench.py --islands $B 2> /dev/null | tail -1 > $O/isl_nomedium_nt.json
timeout 300 python bench.py $B 2> /dev/null | tail -1 > $O/default.json
timeout 300 python bench.py --samples 8 $B 2> /dev/null | tail -1 > $O/default_x8.json
python tools/summ.py $O/isl_medium.json $O/isl_nomedium.json $O/isl_medium124.json $O/isl_medium_ns16.json $O/isl_nomedium_nt.json $O/default.json $O/default_x8.json
