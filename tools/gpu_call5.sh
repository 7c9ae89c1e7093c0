#!/bin/bash
# round 3, GPU call 5: where does the recurrence's time go?  (timing diagnostics: the borders of these runs are wrong on purpose)
set -u
O=gpurun_out/c5
mkdir -p $O
B="--matrix 0 --cpu-seconds 0 --e2e 0 --block-sums 0 --steps 10 --warmup 2"
for diag in 0 1 2 3; do
  WGBSSEG_DP_DIAG=$diag timeout 300 python bench.py $B 2> /dev/null | tail -1 > $O/x32_diag$diag.json
  WGBSSEG_DP_DIAG=$diag WGBSSEG_DP_LEAN=0 timeout 300 python bench.py $B 2> /dev/null | tail -1 > $O/x32_nolean_diag$diag.json
  WGBSSEG_DP_DIAG=$diag timeout 300 python bench.py --sites 3527181 $B 2> /dev/null | tail -1 > $O/eighth_diag$diag.json
  WGBSSEG_DP_DIAG=$diag WGBSSEG_FORCE_STAGES=1 timeout 300 python bench.py --sites 3527181 $B 2> /dev/null | tail -1 > $O/eighth_1stage_diag$diag.json
done
python tools/summ.py $O/*.json
cd tools/micro && ./_build/dp_chain 2>/dev/null | tail -5
