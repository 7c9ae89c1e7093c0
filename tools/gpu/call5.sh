#!/bin/bash
# round 4, call 5: the bench line with the new objects, the N > 1 forms on one GPU, PMC of k_scan with forced carries, the whole GPU suite with durations
set -u
REPO=$PWD
O=$REPO/gpurun_out/r04c5; mkdir -p $O
timeout 900 python bench.py --steps 10 --warmup 3 2> $O/bench.err | tail -1 > $O/bench.json; echo "bench rc $?"; python tools/summ.py $O/bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04c5/bench.json'))
print(json.dumps(d.get('roofline_scan_carries'), indent=0)[:1200])
print('e2e', json.dumps(d.get('end_to_end'))[:600])
print('cpu', json.dumps(d.get('cpu_baseline'))[:400])
PY
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29581 bench.py --gpus 2 --steps 5 --warmup 1 > $O/torchrun2.log 2>&1; echo "torchrun x2 rc $?"; tail -1 $O/torchrun2.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['config']['share_work_max_over_mean'], d.get('matrix'))"
timeout 600 python bench.py --gpus 2 --steps 5 --warmup 1 > $O/group2.log 2>&1; echo "group x2 rc $?"; tail -1 $O/group2.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['config']['share_work_max_over_mean'], d.get('matrix'))"
cd /tmp; export TMPDIR=/tmp
timeout 600 python $REPO/tools/pmc_scan_traffic.py --forced-carries > $O/pmc_scan_carries.log 2>&1; echo "pmc scan carries rc $?"; tail -c 600 $O/pmc_scan_carries.log
cp $REPO/gpurun_out/scan_traffic_carries.json $O/ 2>/dev/null
rm -rf $REPO/gpurun_out/pmc_*
cd $REPO
timeout 1500 python -m pytest tests -q -x -m gpu --durations=25 > $O/gpu_tests_all.log 2>&1; echo "gpu tests: rc $? ($(tail -1 $O/gpu_tests_all.log))"
grep -A 30 "slowest" $O/gpu_tests_all.log | head -40
