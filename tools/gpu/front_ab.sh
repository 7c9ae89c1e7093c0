#!/bin/bash
# Round 6: where the scan pass (k_validate) of a batch should run — from the batch's first event beside the windows pass (default), the same on a
# low-priority stream (WGBSSEG_SCAN_PRIO=1), or behind the windows pass beside the tile plan and the scoring (WGBSSEG_SCAN_AFTER=1): time lines + bench lines.
set -u
O=gpurun_out/${1:-front_ab}; mkdir -p $O
B="--matrix 0 --cpu-seconds 0 --e2e 0 --extras 0 --block-sums 0 --scan-carries 0"
for v in "X=0" "WGBSSEG_SCAN_PRIO=1" "WGBSSEG_SCAN_AFTER=1" "WGBSSEG_SCAN_AFTER=1 WGBSSEG_SCAN_PRIO=1"; do
  tag=$(echo "$v" | tr ' =' '__')
  echo "=== $v"
  for a in "" "--samples 8" "--samples 200" "--sites 3527181"; do
    t2=$(echo "x$a" | tr -d ' -')
    env $v WGBSSEG_PROFILE=2 timeout 300 python bench.py $B --steps 8 --warmup 3 $a > $O/${tag}_$t2.json 2> $O/${tag}_$t2.err
    python tools/summ.py $O/${tag}_$t2.json | cut -c36-; grep "batch of" $O/${tag}_$t2.err | grep -v "batch of 1[0-9][0-9] chunks" | tail -1 | cut -c1-330
  done
done
