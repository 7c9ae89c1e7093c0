#!/bin/bash
# round 3, GPU call 1: micro rates, the new fuzz tests (also on the library with the round-1 carry defect put back), the new bench line, configs[4] at full size
set -u
O=gpurun_out/c1
mkdir -p $O
tools/micro/_build/valu_rates > $O/valu_rates.log 2>&1; echo "valu_rates rc $?"
WGBSSEG_FUZZ_SECONDS=45 WGBSSEG_LIB=$PWD/tools/micro/_build/libwgbsseg_carrybug.so timeout 600 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -q -m gpu -k "aligned_fuzz or test_13" > $O/fuzz_on_carrybug_lib.log 2>&1; echo "fuzz on the carry-bug library: rc $? ($(tail -1 $O/fuzz_on_carrybug_lib.log))"
timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -s > $O/fuzz.log 2>&1; echo "fuzz tests: rc $? ($(tail -1 $O/fuzz.log))"
timeout 900 python bench.py 2> $O/bench.err | tail -1 > $O/bench.json; echo "bench: $(cut -c1-300 $O/bench.json)"; python tools/summ.py $O/bench.json
timeout 300 python bench.py --islands --matrix 0 --cpu-seconds 0 --e2e 0 2> $O/bench_islands.err | tail -1 > $O/bench_islands.json; python tools/summ.py $O/bench_islands.json
WGBSSEG_DEEP_ORACLE_CHUNKS=8 timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -s -k deep_full > $O/deep_full.log 2>&1; echo "deep full: rc $? ($(tail -1 $O/deep_full.log))"
