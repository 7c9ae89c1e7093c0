#!/bin/bash
set -u
O=gpurun_out/c8; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "test_05 or test_06 or test_07 or test_09 or test_13" > $O/parity_subset.log 2>&1; echo "parity subset: rc $? ($(tail -1 $O/parity_subset.log))"
for pair in 0 1; do
  WGBSSEG_DP_PAIR=$pair timeout 300 python tools/dp_timing.py 480 8 2>&1 | grep -v amdgpu.ids | tail -2 > $O/dp_timing_pair$pair.txt; echo "pair=$pair:"; cat $O/dp_timing_pair$pair.txt
  WGBSSEG_DP_PAIR=$pair timeout 300 python tools/dp_timing.py 240 8 2>&1 | grep -v amdgpu.ids | tail -2 > $O/dp_timing240_pair$pair.txt; cat $O/dp_timing240_pair$pair.txt
done
B="--matrix 0 --cpu-seconds 0 --e2e 0 --block-sums 0 --steps 10 --warmup 2"
for pair in 0 1; do
  WGBSSEG_DP_PAIR=$pair timeout 300 python bench.py $B 2> /dev/null | tail -1 > $O/x32_pair$pair.json
  WGBSSEG_DP_PAIR=$pair timeout 300 python bench.py --samples 8 $B 2> /dev/null | tail -1 > $O/x8_pair$pair.json
  WGBSSEG_DP_PAIR=$pair timeout 300 python bench.py --sites 3527181 $B 2> /dev/null | tail -1 > $O/eighth_pair$pair.json
  WGBSSEG_DP_PAIR=$pair timeout 300 python bench.py --islands $B 2> /dev/null | tail -1 > $O/isl_pair$pair.json
done
python tools/summ.py $O/x32_pair*.json $O/x8_pair*.json $O/eighth_pair*.json $O/isl_pair*.json
