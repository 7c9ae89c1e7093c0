#!/bin/bash
# round 4, after the final measurements (same sources): configs[4] at full size with FOUR full 50,000-site chunks against the oracle, a longer deep-mode fuzz, 32-start tiles at x32
set -u
O=gpurun_out/deep_verify; mkdir -p $O
bash tools/gpu/ab.sh deep_verify "main main@WGBSSEG_TI=32" "--samples 32;--sites 3527181"
WGBSSEG_DEEP_FUZZ_SECONDS=170 timeout 400 python -m pytest tests/test_gpu_fuzz.py -q -x -m gpu -k deep_fuzz -s > $O/deep_fuzz_long.log 2>&1; echo "deep fuzz: rc $? $(grep 'deep fuzz:' $O/deep_fuzz_long.log) $(tail -1 $O/deep_fuzz_long.log)"
WGBSSEG_DEEP_ORACLE_CHUNKS=4 timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -x -m gpu -k deep_full_genome -s > $O/deep_full_genome.log 2>&1; echo "deep full genome, 4 full chunks vs oracle: rc $? $(tail -1 $O/deep_full_genome.log)"
cp gpurun_out/deep_full_timing.json $O/ 2>/dev/null
grep "deep full genome" $O/deep_full_genome.log
