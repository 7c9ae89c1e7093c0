#!/bin/bash
# one run each on one box: the committed k_scan (staged carries) and the three variants of tools/experiments/r04_scan_variants_on_staged.patch
set -u
O=gpurun_out/c27; mkdir -p $O
B="--cpu-seconds 0 --e2e 0 --block-sums 0 --matrix 0 --steps 8 --warmup 2 --islands"
for v in committed stagedDOT stagedAHEAD2 stagedAHEAD2DOT; do
  L=$PWD/wgbs_tools_amd/csrc/libwgbsseg.so; [ $v != committed ] && L=$PWD/tools/micro/_build/libwgbsseg_$v.so
  WGBSSEG_LIB=$L timeout 14 python bench.py $B 2> /dev/null | tail -1 > $O/isl_$v.json
  python tools/summ.py $O/isl_$v.json
done
