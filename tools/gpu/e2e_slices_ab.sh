#!/bin/bash
# Round 6: `wgbstools segment` end to end with the regions in slices (BED rows of slice k written while slice k + 1 is segmented) against the one-piece form.
#   bash tools/gpu/e2e_slices_ab.sh [samples ...]   -> gpurun_out/e2e_slices/
set -u
O=gpurun_out/e2e_slices; mkdir -p $O
for N in ${@:-32 200}; do
  for n in 1 4 2 8 1 4; do
    WGBSSEG_BED_SLICES=$n timeout 900 python tools/e2e_bench.py --samples $N --keep > $O/e2e_x${N}_slices$n.log 2>&1
    echo "x$N slices $n: rc $? | $(grep '^run [12]' $O/e2e_x${N}_slices$n.log | sed 's/rc 0, //; s/CpG-sites.s end to end; \[wt segment\] found [0-9,]* blocks | \[wt segment\] phases: //' | cut -c1-260 | tr '\n' '|')"
    grep BED $O/e2e_x${N}_slices$n.log
  done
  rm -rf /tmp/wgbs_e2e
done
