#!/bin/bash
set -u
O=gpurun_out/c9; mkdir -p $O
B="--matrix 0 --cpu-seconds 0 --e2e 0 --block-sums 0 --steps 10 --warmup 2"
for v in main norfl main norfl; do
  if [ $v = norfl ]; then export WGBSSEG_LIB=$PWD/tools/micro/_build/libwgbsseg_norfl.so; else unset WGBSSEG_LIB; fi
  timeout 300 python bench.py $B 2> /dev/null | tail -1 > $O/x32_$v.json
  timeout 300 python bench.py --islands $B 2> /dev/null | tail -1 > $O/isl_$v.json
  timeout 300 python bench.py --samples 8 $B 2> /dev/null | tail -1 > $O/x8_$v.json
  timeout 300 python bench.py --sites 3527181 $B 2> /dev/null | tail -1 > $O/eighth_$v.json
  python tools/summ.py $O/x32_$v.json $O/isl_$v.json $O/x8_$v.json $O/eighth_$v.json
done
