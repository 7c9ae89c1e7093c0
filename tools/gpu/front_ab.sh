#!/bin/bash
# Round 6: where the scan pass (k_validate, on the lowest-priority stream) of a batch should start — with the batch, beside the windows pass (WGBSSEG_SCAN_AFTER=0),
# behind the windows pass (1), or behind the tile plan, beside the first scoring tiles only (2): time lines + bench lines, three alternating repetitions.
set -u
O=gpurun_out/${1:-front_ab}; mkdir -p $O
B="--matrix 0 --cpu-seconds 0 --e2e 0 --extras 0 --block-sums 0 --scan-carries 0"
for rep in 1 2 3; do
  for m in 0 1 2; do
    for a in "" "--samples 8" "--samples 200" "--sites 3527181"; do
      t2=$(echo "m${m}_r${rep}_x$a" | tr -d ' -')
      WGBSSEG_SCAN_AFTER=$m WGBSSEG_PROFILE=2 timeout 300 python bench.py $B --steps 8 --warmup 3 $a > $O/$t2.json 2> $O/$t2.err
      echo "after=$m rep $rep [$a] $(python tools/summ.py $O/$t2.json | cut -c36-)"
      [ $rep = 1 ] && grep "batch of" $O/$t2.err | grep -v "batch of 1[0-9][0-9] chunks\|batch of [0-9][0-9] chunks" | tail -1 | cut -c1-330
    done
  done
done
