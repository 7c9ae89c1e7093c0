#!/bin/bash
set -u
O=gpurun_out/c11; mkdir -p $O
B="--matrix 0 --cpu-seconds 0 --e2e 0 --block-sums 0 --steps 10 --warmup 2"
for nw in 0 11 5; do
  WGBSSEG_DP_NW=$nw timeout 300 python bench.py $B 2> /dev/null | tail -1 > $O/x32_nw$nw.json
  WGBSSEG_DP_NW=$nw timeout 300 python bench.py --samples 8 $B 2> /dev/null | tail -1 > $O/x8_nw$nw.json
done
for nw in 0 11 3; do
  WGBSSEG_DP_NW=$nw timeout 300 python bench.py --islands $B 2> /dev/null | tail -1 > $O/isl_nw$nw.json
done
python tools/summ.py $O/x32_nw*.json $O/x8_nw*.json $O/isl_nw*.json
WGBSSEG_DP_NW=11 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "test_17 or test_06 or test_13" > $O/parity_nw11.log 2>&1; echo "parity subset (11 workers): rc $? ($(tail -1 $O/parity_nw11.log))"
