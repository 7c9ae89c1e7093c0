#!/bin/bash
# rocprofv3 kernel statistics of one GPU's share of 8 (gated stages), then the round's bench lines on this box (tools/gpu/bench_lines.sh)
set -u
REPO=$PWD; O=$REPO/gpurun_out/share_prof; mkdir -p $O
B="--cpu-seconds 0 --e2e 0 --extras 0 --matrix 0 --block-sums 0 --scan-carries 0"
(cd /tmp; export TMPDIR=/tmp
 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o stats -- python $REPO/bench.py $B --sites 3527181 --steps 20 --warmup 3 > $O/rocprof_share.log 2>&1; echo "rocprofv3 share: rc $?"
 find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/rocprofv3_kernel_stats_share_of_8_x32.csv; rm -rf $O/prof)
tail -1 $O/rocprof_share.log | python tools/summ.py /dev/stdin | head -1 | cut -c36-
head -12 $O/rocprofv3_kernel_stats_share_of_8_x32.csv | cut -d, -f1-4,6,7 | cut -c1-160
bash tools/gpu/bench_lines.sh
