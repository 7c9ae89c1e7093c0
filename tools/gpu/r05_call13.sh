#!/bin/bash
# round 5, call 13: how much does the block reduction depend on its occupancy?  The product holds 4 workgroups per CU (36 KB of LDS each); padded builds hold 3 and 2.
set -u
REPO=$PWD
O=$REPO/gpurun_out/r05c13; mkdir -p $O
for v in main bspad8k bspad20k main; do
  if [ "$v" = "main" ]; then LIBENV=""; else LIBENV="WGBSSEG_ALLOW_LIB_OVERRIDE=1 WGBSSEG_LIB=$REPO/tools/micro/_build/libwgbsseg_$v.so"; fi
  env $LIBENV timeout 300 python bench.py --steps 3 --warmup 1 --matrix 0 --cpu-seconds 0 --e2e 0 --extras 0 --scan-carries 0 2> /dev/null | tail -1 > $O/bs_$v.json
  python -c "
import json; d=json.load(open('$O/bs_$v.json'))['block_sums']; print('%-12s .bin rows %.4f ms (all %s) = %.3f of peak; means %.4f, raw %.4f' % ('$v', d['ms_bin_rows'], ['%.3f' % x for x in d['ms_bin_rows_all']], d['frac_of_hbm_peak'], d['ms_means'], d['ms_raw_sums']))"
done 2>&1 | tee $O/block_sums_occupancy.txt
