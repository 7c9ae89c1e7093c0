#!/bin/bash
# after adopting the LDS-staged carries: PMC files for the new source state, the islands bench line with its traffic, parity file, fuzz slice
set -u
REPO=$PWD; O=$REPO/gpurun_out/c25; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 50 python $REPO/tools/pmc_scan_traffic.py > $O/pmc_scan.log 2>&1; echo "pmc scan traffic: rc $?"
timeout 50 python $REPO/tools/pmc_scan_traffic.py --islands > $O/pmc_scan_islands.log 2>&1; echo "pmc scan traffic (islands): rc $?"
timeout 40 python $REPO/tools/pmc_cost_sq.py > $O/pmc_cost_sq.log 2>&1; echo "pmc cost sq: rc $?"
cp $REPO/gpurun_out/scan_traffic.json $REPO/gpurun_out/scan_traffic_islands.json $REPO/gpurun_out/pmc_cost_sq.json $O/ 2>/dev/null
cp $REPO/gpurun_out/scan_traffic.json $REPO/profiles/r03_scan_traffic.json; cp $REPO/gpurun_out/scan_traffic_islands.json $REPO/profiles/r03_scan_traffic_islands.json
rm -rf $REPO/gpurun_out/pmc_*
cd $REPO
timeout 40 python bench.py --islands --matrix 0 --cpu-seconds 0 --e2e 0 --steps 10 --warmup 2 2> /dev/null | tail -1 > $O/bench_islands.json; python tools/summ.py $O/bench_islands.json
timeout 90 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > $O/parity.log 2>&1; echo "parity: rc $? ($(tail -1 $O/parity.log))"
WGBSSEG_FUZZ_SECONDS=12 timeout 60 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x -s > $O/fuzz.log 2>&1; echo "fuzz: rc $? ($(tail -1 $O/fuzz.log)) $(grep -h 'aligned fuzz' $O/fuzz.log | tail -1)"
