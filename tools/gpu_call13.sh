#!/bin/bash
# A/B of the carry spacing of k_scan (128 sites by default; builds of tools/build_carry_libs.sh with 256 / 512 / 1024)
set -u
O=gpurun_out/c13; mkdir -p $O
B="--cpu-seconds 0 --e2e 0 --block-sums 0 --matrix 0"
for s in 7 8 9 10; do
  L=$PWD/wgbs_tools_amd/csrc/libwgbsseg.so; [ $s != 7 ] && L=$PWD/tools/micro/_build/libwgbsseg_carry$s.so
  WGBSSEG_LIB=$L timeout 300 python bench.py --islands $B --steps 10 --warmup 2 2> /dev/null | tail -1 > $O/isl_carry$s.json
  WGBSSEG_LIB=$L timeout 300 python bench.py --sites 3527181 --samples 64 --max-cpg 5000 --max-bp 1000000 --chunk 50000 $B --steps 2 --warmup 1 2> $O/deep_carry$s.err | tail -1 > $O/deep_carry$s.json
done
python tools/summ.py $O/isl_carry*.json $O/deep_carry*.json
for s in 9 10; do
  L=$PWD/tools/micro/_build/libwgbsseg_carry$s.so
  WGBSSEG_LIB=$L timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > $O/parity_carry$s.log 2>&1; echo "parity carry$s: rc $? ($(tail -1 $O/parity_carry$s.log))"
  WGBSSEG_LIB=$L WGBSSEG_FUZZ_SECONDS=30 timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x -s > $O/fuzz_carry$s.log 2>&1; echo "fuzz carry$s: rc $? ($(tail -1 $O/fuzz_carry$s.log)) $(grep -h 'aligned fuzz' $O/fuzz_carry$s.log | tail -1)"
done
