#!/bin/bash
# round 5, call 3: barrier waits and load-issue time of EVERY role of k_dp (which worker is late?), and the first run of bench.py's `extras`
set -u
O=$PWD/gpurun_out/r05c3; mkdir -p $O
for nch in 480 240; do timeout 200 python tools/dp_timing.py $nch 8 2>&1 | grep -v "WGBSSEG_LIB\|amdgpu.ids" | tail -16 | cut -c1-300 > $O/dp_roles_$nch.txt; done
cat $O/dp_roles_480.txt | tail -14
timeout 600 python bench.py --steps 3 --warmup 1 --matrix 0 --cpu-seconds 0 --e2e 0 --block-sums 0 --scan-carries 0 2> $O/bench_extras.err | tail -1 > $O/bench_extras.json
python -c "
import json; d=json.load(open('$O/bench_extras.json')); print(json.dumps(d.get('extras'), indent=1)[:6000])"
tail -5 $O/bench_extras.err
