#!/bin/bash
# rocprofv3 kernel statistics of the OTHER rows of the metric: x8, x200 and the deep configuration (x512, max_cpg 5000, max_bp 1e6,
# chunk 50000), besides final_a.sh's x32 / x32-islands.    R=r06 bash tools/gpu/rocprof_rows.sh   -> gpurun_out/rows/ (+ profiles/ of the box's copy)
set -u
R=${R:-r06}
REPO=$PWD; O=$REPO/gpurun_out/rows; mkdir -p $O
B="--cpu-seconds 0 --e2e 0 --extras 0 --matrix 0 --block-sums 0 --scan-carries 0"
cd /tmp; export TMPDIR=/tmp
run() {  # name, bench args
  local name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o stats -- python $REPO/bench.py $B "$@" > $O/rocprof_$name.log 2>&1; echo "rocprofv3 $name: rc $?"
  find $O/prof_$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/rocprofv3_kernel_stats_$name.csv; rm -rf $O/prof_$name
  tail -1 $O/rocprof_$name.log | python $REPO/tools/summ.py /dev/stdin | head -1 | cut -c36-
  head -6 $O/rocprofv3_kernel_stats_$name.csv | cut -d, -f1-4 | cut -c1-140
}
run 28M_x8 --samples 8 --steps 5 --warmup 2
run 28M_x200 --samples 200 --steps 3 --warmup 1
run 28M_x512_deep --samples 512 --max-cpg 5000 --max-bp 1000000 --chunk 50000 --steps 1 --warmup 0
