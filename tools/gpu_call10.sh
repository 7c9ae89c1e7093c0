#!/bin/bash
set -u
O=gpurun_out/c10; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "test_17 or test_16 or test_05 or test_06 or test_07 or test_09 or test_13" > $O/parity_subset.log 2>&1; echo "parity subset: rc $? ($(tail -1 $O/parity_subset.log))"; grep -E "^E|Error" $O/parity_subset.log | head -8
WGBSSEG_FUZZ_SECONDS=40 timeout 300 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -s -k "time_boxed" > $O/fuzz.log 2>&1; echo "fuzz: rc $? ($(tail -1 $O/fuzz.log)) $(grep 'aligned fuzz' $O/fuzz.log)"
B="--matrix 0 --cpu-seconds 0 --e2e 0 --block-sums 0 --steps 10 --warmup 2"
for w in 1 0; do for nw in 0 7; do
  WGBSSEG_DP_NW=$nw WGBSSEG_DP_WLEAN=$w timeout 300 python bench.py --islands $B 2> /dev/null | tail -1 > $O/isl_wlean${w}_nw$nw.json
  python tools/summ.py $O/isl_wlean${w}_nw$nw.json
done; done
