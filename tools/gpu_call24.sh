#!/bin/bash
# A/B: the row's carries staged in LDS, written when the row is done (-DWG_SCAN_LDSCARRY build)
set -u
O=gpurun_out/c24; mkdir -p $O
B="--cpu-seconds 0 --e2e 0 --block-sums 0 --matrix 0 --steps 10 --warmup 2"
for v in default scanLDS default scanLDS; do
  L=$PWD/wgbs_tools_amd/csrc/libwgbsseg.so; [ $v != default ] && L=$PWD/tools/micro/_build/libwgbsseg_$v.so
  WGBSSEG_LIB=$L timeout 300 python bench.py --islands $B 2> /dev/null | tail -1 > $O/isl_$v.json
  python tools/summ.py $O/isl_$v.json
done
L=$PWD/tools/micro/_build/libwgbsseg_scanLDS.so
WGBSSEG_LIB=$L timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "test_03 or test_04 or test_13 or test_16" > $O/parity.log 2>&1; echo "parity subset: rc $? ($(tail -1 $O/parity.log))"
WGBSSEG_LIB=$L WGBSSEG_FUZZ_SECONDS=10 timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x -s > $O/fuzz.log 2>&1; echo "fuzz: rc $? ($(tail -1 $O/fuzz.log)) $(grep -h 'aligned fuzz' $O/fuzz.log | tail -1)"
