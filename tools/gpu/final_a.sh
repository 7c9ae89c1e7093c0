#!/bin/bash
# Final measurements on the final sources, part A (R=rNN names the files): the whole GPU suite, the bench lines, rocprofv3 kernel statistics, the PMC passes
# (HBM traffic of k_cost / of the scan pass without and with forced carries and with islands; SQ counters of k_cost).  Everything lands in gpurun_out/final/.
set -u
R=${R:-r06}
REPO=$PWD
O=$REPO/gpurun_out/final
mkdir -p $O
if [ "${SUITE:-1}" = "1" ]; then
timeout 1500 python -m pytest tests -q -x -m gpu --durations=12 > $O/gpu_tests_all.log 2>&1; echo "gpu tests: rc $? ($(tail -1 $O/gpu_tests_all.log))"
else      # (SUITE=0: after a HOST-side change — the device code of the suite's run stands: the tests that go through the changed host code only)
timeout 600 python -m pytest tests/test_gpu_driver.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -q -x -m gpu -k "driver or x32 or x200 or test_09 or test_19d or test_08" > $O/gpu_tests_host_subset.log 2>&1; echo "gpu tests (host-side subset): rc $? ($(tail -1 $O/gpu_tests_host_subset.log))"
fi
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o stats -- python $REPO/bench.py --steps 5 --warmup 2 --cpu-seconds 0 --e2e 0 --extras 0 --matrix 0 > $O/rocprof_bench.log 2>&1; echo "rocprofv3 stats: rc $?"
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/rocprofv3_kernel_stats_28M_x32.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_isl -o stats -- python $REPO/bench.py --islands --steps 5 --warmup 2 --cpu-seconds 0 --e2e 0 --extras 0 --matrix 0 > $O/rocprof_bench_islands.log 2>&1
find $O/prof_isl -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/rocprofv3_kernel_stats_28M_x32_islands.csv
timeout 900 python $REPO/tools/pmc_cost_traffic.py > $O/pmc_cost_traffic.log 2>&1; echo "pmc cost traffic: rc $?"
timeout 600 python $REPO/tools/pmc_scan_traffic.py > $O/pmc_scan.log 2>&1; echo "pmc scan traffic: rc $?"
timeout 600 python $REPO/tools/pmc_scan_traffic.py --forced-carries > $O/pmc_scan_carries.log 2>&1; echo "pmc scan traffic (forced carries): rc $?"
timeout 600 python $REPO/tools/pmc_scan_traffic.py --islands > $O/pmc_scan_islands.log 2>&1; echo "pmc scan traffic (islands): rc $?"
timeout 600 python $REPO/tools/pmc_cost_sq.py > $O/pmc_cost_sq.log 2>&1; echo "pmc cost sq: rc $?"
for f in cost_traffic scan_traffic scan_traffic_carries scan_traffic_islands pmc_cost_sq; do cp $REPO/gpurun_out/$f.json $O/ 2>/dev/null; cp $REPO/gpurun_out/$f.json $REPO/profiles/${R}_$f.json 2>/dev/null; done
rm -rf $O/prof $O/prof_isl $REPO/gpurun_out/pmc_*
cd $REPO
# (the PMC files above are now under profiles/ of THIS box's copy: the bench line below reports them; they come home under gpurun_out/final/)
timeout 900 python bench.py --steps 20 --warmup 5 2> $O/bench.err | tail -1 > $O/bench.json; echo "bench: $(cut -c1-160 $O/bench.json)"; python tools/summ.py $O/bench.json
SECONDS=0; timeout 900 python bench.py 2> $O/bench_default_flags.err | tail -1 > $O/bench_default_flags.json; echo "bench with the driver's default flags: $SECONDS s"; python tools/summ.py $O/bench_default_flags.json | head -2
B="--matrix 0 --cpu-seconds 0 --e2e 0 --extras 0 --scan-carries 0 --steps 10 --warmup 2"
timeout 300 python bench.py --islands $B 2> /dev/null | tail -1 > $O/bench_islands.json
timeout 300 python bench.py --islands --samples 8 $B 2> /dev/null | tail -1 > $O/bench_islands_x8.json
timeout 300 python bench.py --sites 3527181 $B 2> /dev/null | tail -1 > $O/bench_one_eighth.json
python tools/summ.py $O/bench_islands.json $O/bench_islands_x8.json $O/bench_one_eighth.json
