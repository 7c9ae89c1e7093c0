#!/bin/bash
# Final measurements, part C: BASELINE.json configs[4] at full size (28 M x 512, max_cpg 5000) with N full 50,000-site chunks against the oracle's restatement
# (two minutes of all host threads each), and a longer deep-mode fuzz.   N=${1:-8}
set -u
N=${1:-8}
O=$PWD/gpurun_out/final; mkdir -p $O
WGBSSEG_DEEP_FUZZ_SECONDS=120 timeout 300 python -m pytest tests/test_gpu_fuzz.py -q -x -m gpu -k deep_fuzz -s > $O/deep_fuzz_long.log 2>&1; echo "deep fuzz: rc $? $(grep 'deep fuzz:' $O/deep_fuzz_long.log) $(tail -1 $O/deep_fuzz_long.log)"
WGBSSEG_DEEP_ORACLE_CHUNKS=$N timeout 2400 python -m pytest tests/test_gpu_fullsize.py -q -x -m gpu -k deep_full_genome -s > $O/deep_full_genome.log 2>&1; echo "deep full genome, $N full chunks vs oracle: rc $? $(tail -1 $O/deep_full_genome.log)"
cp gpurun_out/deep_full_timing.json $O/ 2>/dev/null
grep "deep full genome" $O/deep_full_genome.log
