#!/bin/bash
# Final measurements, part A2: what needs the PMC / instruction-mix files of THIS source state under profiles/ before it runs — the PMC traffic of k_cost (needs
# tools/micro/_build/fetch_calib, built before the call), then the bench lines (20 steps; the driver's flags; islands; x8 islands; one eighth) with everything
# attached, and the time lines.  gpurun_out/final/.
set -u
R=${R:-r06}
REPO=$PWD; O=$REPO/gpurun_out/final; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 900 python $REPO/tools/pmc_cost_traffic.py > $O/pmc_cost_traffic.log 2>&1; echo "pmc cost traffic: rc $?"
cp $REPO/gpurun_out/cost_traffic.json $O/ 2>/dev/null; cp $REPO/gpurun_out/cost_traffic.json $REPO/profiles/${R}_cost_traffic.json 2>/dev/null
rm -rf $REPO/gpurun_out/pmc_*
cd $REPO
timeout 900 python bench.py --steps 20 --warmup 5 2> $O/bench.err | tail -1 > $O/bench.json; echo "bench: $(cut -c1-160 $O/bench.json)"; python tools/summ.py $O/bench.json
SECONDS=0; timeout 900 python bench.py 2> $O/bench_default_flags.err | tail -1 > $O/bench_default_flags.json; echo "bench with the driver's default flags: $SECONDS s"; python tools/summ.py $O/bench_default_flags.json | head -2
B="--matrix 0 --cpu-seconds 0 --e2e 0 --extras 0 --scan-carries 0 --steps 10 --warmup 2"
timeout 300 python bench.py --sites 3527181 $B 2> /dev/null | tail -1 > $O/bench_one_eighth.json; python tools/summ.py $O/bench_one_eighth.json
bash tools/gpu/timeline.sh > $O/timelines.txt 2>&1; tail -30 $O/timelines.txt | cut -c1-400
