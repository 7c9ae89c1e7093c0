#!/bin/bash
# round 4, final measurements on the final sources: rocprofv3 kernel statistics, PMC passes (HBM traffic of k_cost, of the scan pass without / with forced carries / with
# islands; SQ counters of k_cost), the bench lines, both N > 1 forms on one GPU, a long fuzz on new seeds, the whole GPU suite.  Everything lands in gpurun_out/final/.
set -u
REPO=$PWD
O=$REPO/gpurun_out/final
mkdir -p $O
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
for f in cost_traffic scan_traffic scan_traffic_carries scan_traffic_islands pmc_cost_sq; do cp $REPO/gpurun_out/$f.json $O/ 2>/dev/null; cp $REPO/gpurun_out/$f.json $REPO/profiles/r04_$f.json 2>/dev/null; done
rm -rf $O/prof $O/prof_isl $REPO/gpurun_out/pmc_*
cd $REPO
# (the PMC files above are now under profiles/ of THIS box's copy: the bench line below reports them; they come home under gpurun_out/final/)
timeout 900 python bench.py --steps 20 --warmup 5 2> $O/bench.err | tail -1 > $O/bench.json; echo "bench: $(cut -c1-160 $O/bench.json)"; python tools/summ.py $O/bench.json
B="--matrix 0 --cpu-seconds 0 --e2e 0 --extras 0 --scan-carries 0 --steps 10 --warmup 2"
timeout 300 python bench.py --islands $B 2> /dev/null | tail -1 > $O/bench_islands.json
timeout 300 python bench.py --islands --samples 8 $B 2> /dev/null | tail -1 > $O/bench_islands_x8.json
timeout 300 python bench.py --sites 3527181 $B 2> /dev/null | tail -1 > $O/bench_one_eighth.json
python tools/summ.py $O/bench_islands.json $O/bench_islands_x8.json $O/bench_one_eighth.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29579 bench.py --gpus 2 --steps 5 --warmup 1 > $O/torchrun2.log 2>&1; echo "torchrun x2: rc $?"
WGBSSEG_BENCH_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29580 bench.py --gpus 1 --steps 5 --warmup 1 --matrix 0 > $O/torchrun1_rccl.log 2>&1; echo "torchrun x1 (RCCL group): rc $?"
timeout 600 python bench.py --gpus 8 --steps 5 --warmup 1 > $O/group8_on_one_gpu.log 2>&1; echo "group of 8 shares on one GPU: rc $? $(tail -1 $O/group8_on_one_gpu.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['config']['share_work_max_over_mean'], (d.get('matrix') or {}).get('rows'))")"
if [ "${FINAL_FUZZ:-1}" = "1" ]; then      # (FINAL_FUZZ=0: a re-run after a host-side change — the kernels' fuzz log of the round stands)
timeout 400 python tools/aligned_fuzz.py 4400000 1000000 200 > $O/fuzz_long_aligned.log 2>&1; echo "aligned fuzz: $(tail -1 $O/fuzz_long_aligned.log)"
timeout 200 python tools/extra_fuzz.py 40000 100000 90 > $O/fuzz_long_uniform.log 2>&1; echo "uniform fuzz: $(tail -1 $O/fuzz_long_uniform.log)"
fi
timeout 1500 python -m pytest tests -q -x -m gpu --durations=12 > $O/gpu_tests_all.log 2>&1; echo "gpu tests: rc $? ($(tail -1 $O/gpu_tests_all.log))"
