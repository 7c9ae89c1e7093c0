#!/bin/bash
# one GPU's share of 8: how the scoring time depends on the number of stages (drain per launch vs the recurrence beside it)
set -u
O=gpurun_out/share_stage_sweep; mkdir -p $O
B="--matrix 0 --cpu-seconds 0 --e2e 0 --extras 0 --block-sums 0 --scan-carries 0 --steps 10 --warmup 3 --sites 3527181"
for s in 1 2 4 6 8 10 12; do
  WGBSSEG_FORCE_STAGES=$s timeout 300 python bench.py $B 2>/dev/null | tail -1 > $O/st$s.json; echo "stages $s: $(python tools/summ.py $O/st$s.json | cut -c36-)"
done
