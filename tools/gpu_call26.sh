#!/bin/bash
# paired A/B on one box: the library before the LDS-staged carries (prev) and the committed one
set -u
O=gpurun_out/c26; mkdir -p $O
B="--cpu-seconds 0 --e2e 0 --block-sums 0 --matrix 0 --steps 10 --warmup 2 --islands"
for v in prev staged prev staged; do
  L=$PWD/wgbs_tools_amd/csrc/libwgbsseg.so; [ $v = prev ] && L=$PWD/tools/micro/_build/libwgbsseg_prev.so
  WGBSSEG_LIB=$L timeout 20 python bench.py $B 2> /dev/null | tail -1 > $O/isl_$v.json
  python tools/summ.py $O/isl_$v.json
done
