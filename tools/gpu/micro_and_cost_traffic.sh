#!/bin/bash
# does clustering cheap VALU instructions pay (micro), the baseline of this box, HBM traffic of k_cost by PMC
set -u
REPO=$PWD
O=$REPO/gpurun_out/micro
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 120 $REPO/tools/micro/_build/valu_cluster > $O/valu_cluster.json 2> $O/valu_cluster.err; echo "valu_cluster rc $?"
cd $REPO
timeout 400 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --e2e 0 --extras 0 2> $O/bench_base.err | tail -1 > $O/bench_base.json; echo "bench rc $?"; python tools/summ.py $O/bench_base.json
cd /tmp
timeout 900 python $REPO/tools/pmc_cost_traffic.py > $O/pmc_cost_traffic.log 2>&1; echo "pmc cost traffic rc $?"
cp $REPO/gpurun_out/cost_traffic.json $O/ 2>/dev/null
rm -rf $REPO/gpurun_out/pmc_ct_*
tail -c 1500 $O/pmc_cost_traffic.log
