#!/bin/bash
# A mid-round check of the library in the tree on ONE box: the parity tests that go through the batch path (every tile class, the plain path, the
# native stitcher, the fuzz slices), then the time lines of the whole-genome step and of one GPU's share of 8, then short bench lines.
#   [PROF=1] bash tools/gpu/check.sh [OUT] [pytest -k expression]        -> gpurun_out/<OUT>/      (PROF=1: rocprofv3 kernel statistics of a short bench as well)
set -u
O=gpurun_out/${1:-check}; mkdir -p $O
K=${2:-}
B="--matrix 0 --cpu-seconds 0 --e2e 0 --extras 0 --block-sums 0 --scan-carries 0"
if [ -n "$K" ]; then
  timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_driver.py tests/test_gpu_segmentor_bin.py tests/test_gpu_integration_stub.py -x -q -m gpu -k "$K" > $O/pytest.log 2>&1
else
  timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_driver.py tests/test_gpu_segmentor_bin.py tests/test_gpu_integration_stub.py -x -q -m gpu > $O/pytest.log 2>&1
fi
echo "parity: rc $? ($(tail -1 $O/pytest.log))"; grep -E "^(FAILED|ERROR)|Error|assert" $O/pytest.log | head -20
WGBSSEG_PROFILE=2 WGBSSEG_PROFILE_STITCH=1 timeout 300 python bench.py $B --steps 4 --warmup 2 > $O/full.json 2> $O/full.err; python tools/summ.py $O/full.json | cut -c36-; grep "batch of\|\[stitch\]" $O/full.err | tail -14
WGBSSEG_PROFILE=2 WGBSSEG_PROFILE_STITCH=1 timeout 300 python bench.py $B --steps 4 --warmup 2 --sites 3527181 > $O/share.json 2> $O/share.err; python tools/summ.py $O/share.json | cut -c36-; grep "batch of\|\[stitch\]" $O/share.err | tail -14
if [ "${PROF:-0}" = "1" ]; then
  R=$PWD; (cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o stats -- python $R/bench.py $B --steps 5 --warmup 2 > $R/$O/rocprof.log 2>&1)
  find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; rm -rf $O/prof
  cut -d, -f1-4,6,7 $O/kernel_stats.csv | cut -c1-150 | head -24
fi
for a in "" "--samples 8" "--islands" "--sites 3527181"; do
  tag=$(echo "x$a" | tr -d ' -'); timeout 300 python bench.py $B --steps 10 --warmup 3 $a 2> $O/bench_$tag.err | tail -1 > $O/bench_$tag.json
  echo "== bench $a"; python tools/summ.py $O/bench_$tag.json 2>&1 | cut -c36-
done
