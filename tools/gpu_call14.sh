#!/bin/bash
# (no source change) one GPU's share of 8 / 4 / 2: stage counts x recurrence kernel beside the scoring (WGBSSEG_DP16=0: the full-LDS k_dp in every stage),
# then a long slice of the aligned fuzz and the reference-binary fuzz on the committed library
set -u
O=gpurun_out/c14; mkdir -p $O
B="--cpu-seconds 0 --e2e 0 --block-sums 0 --matrix 0 --steps 20 --warmup 5"
for st in 4 8 12 16; do
  WGBSSEG_MIN_STAGES=$st timeout 300 python bench.py --sites 3527181 $B 2> /dev/null | tail -1 > $O/eighth_dp16_s$st.json
  WGBSSEG_DP16=0 WGBSSEG_MIN_STAGES=$st timeout 300 python bench.py --sites 3527181 $B 2> /dev/null | tail -1 > $O/eighth_fulldp_s$st.json
done
timeout 300 python bench.py --sites 7054362 $B 2> /dev/null | tail -1 > $O/quarter_default.json
WGBSSEG_MIN_STAGES=8 timeout 300 python bench.py --sites 7054362 $B 2> /dev/null | tail -1 > $O/quarter_dp16_s8.json
WGBSSEG_DP16=0 WGBSSEG_MIN_STAGES=8 timeout 300 python bench.py --sites 7054362 $B 2> /dev/null | tail -1 > $O/quarter_fulldp_s8.json
timeout 300 python bench.py --sites 14108724 $B 2> /dev/null | tail -1 > $O/half_default.json
WGBSSEG_MIN_STAGES=8 timeout 300 python bench.py --sites 14108724 $B 2> /dev/null | tail -1 > $O/half_dp16_s8.json
python tools/summ.py $O/eighth_*.json $O/quarter_*.json $O/half_*.json
WGBSSEG_FUZZ_SECONDS=240 WGBSSEG_FUZZ_FIRST=700000 timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x -s > $O/fuzz_long.log 2>&1; echo "long fuzz: rc $? ($(tail -1 $O/fuzz_long.log)) $(grep -h 'aligned fuzz' $O/fuzz_long.log | tail -1)"
