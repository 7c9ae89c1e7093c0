#!/bin/bash
# the multi-process bench path with ONE rank on RCCL (init with device_id, host group, /dev/shm slots, device barrier, all_reduce) and the default bench line again
set -u
O=gpurun_out/c15; mkdir -p $O
WGBSSEG_BENCH_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29581 bench.py --gpus 1 --steps 10 --warmup 2 --cpu-seconds 0 --e2e 0 --matrix 0 > $O/dist1.log 2>&1; echo "one rank on RCCL: rc $?"; grep '^{' $O/dist1.log | tail -1 > $O/dist1.json; python tools/summ.py $O/dist1.json; grep -v '^{' $O/dist1.log | tail -5
timeout 900 python bench.py --steps 20 --warmup 5 2> $O/bench.err | tail -1 > $O/bench.json; python tools/summ.py $O/bench.json; python - <<'PY'
import json
d = json.loads(open('gpurun_out/c15/bench.json').read())
print('traffic', d['roofline_scan'].get('traffic'), 'sq', {k: d['roofline'].get(k) for k in ('lds_conflict_frac', 'pmc')})
PY
