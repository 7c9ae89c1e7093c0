#!/bin/bash
# A/B: two iterations in flight per wavefront in the scan pass with carries (-DWG_SCAN_AHEAD2), alone and with the byte dot products
set -u
O=gpurun_out/c20; mkdir -p $O
B="--cpu-seconds 0 --e2e 0 --block-sums 0 --matrix 0 --steps 10 --warmup 2"
for v in default scanAHEAD2 scanAHEAD2DOT default scanAHEAD2 scanAHEAD2DOT; do
  L=$PWD/wgbs_tools_amd/csrc/libwgbsseg.so; [ $v != default ] && L=$PWD/tools/micro/_build/libwgbsseg_$v.so
  WGBSSEG_LIB=$L timeout 300 python bench.py --islands $B 2> /dev/null | tail -1 > $O/isl_$v.json
  python tools/summ.py $O/isl_$v.json
done
for v in scanAHEAD2 scanAHEAD2DOT; do
L=$PWD/tools/micro/_build/libwgbsseg_$v.so
WGBSSEG_LIB=$L timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "test_03 or test_04 or test_13 or test_11 or test_16" > $O/parity_$v.log 2>&1; echo "$v parity subset: rc $? ($(tail -1 $O/parity_$v.log))"
WGBSSEG_LIB=$L WGBSSEG_FUZZ_SECONDS=15 timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x -s > $O/fuzz_$v.log 2>&1; echo "$v fuzz: rc $? ($(tail -1 $O/fuzz_$v.log)) $(grep -h 'aligned fuzz' $O/fuzz_$v.log | tail -1)"
done
