#!/bin/bash
# A/B of the second scoring stream (WGBSSEG_COST_STREAMS=1: off)
set -u
O=gpurun_out/c12; mkdir -p $O
B="--cpu-seconds 0 --e2e 0 --block-sums 0 --steps 10 --warmup 2"
for cs in 2 1; do
  WGBSSEG_COST_STREAMS=$cs timeout 300 python bench.py --sites 3527181 --matrix 0 $B 2> /dev/null | tail -1 > $O/eighth_cs$cs.json
  WGBSSEG_COST_STREAMS=$cs timeout 300 python bench.py --islands --matrix 0 $B 2> /dev/null | tail -1 > $O/isl_cs$cs.json
  WGBSSEG_COST_STREAMS=$cs timeout 300 python bench.py --islands --samples 8 --matrix 0 $B 2> /dev/null | tail -1 > $O/islx8_cs$cs.json
  WGBSSEG_COST_STREAMS=$cs timeout 600 python bench.py --matrix 1 $B 2> /dev/null | tail -1 > $O/x32_cs$cs.json
done
python tools/summ.py $O/eighth_cs*.json $O/isl_cs*.json $O/islx8_cs*.json $O/x32_cs*.json
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > $O/parity.log 2>&1; echo "parity: rc $? ($(tail -1 $O/parity.log))"
WGBSSEG_FUZZ_SECONDS=40 timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x -s > $O/fuzz.log 2>&1; echo "fuzz: rc $? ($(tail -1 $O/fuzz.log)) $(grep -h 'aligned fuzz' $O/fuzz.log | tail -1)"
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -k "not deep_full" > $O/fullsize.log 2>&1; echo "fullsize: rc $? ($(tail -1 $O/fullsize.log))"
