#!/bin/bash
# round 5, call 5: the counting kernel of pat2beta, round 4's (one thread per byte) against round 5's (tiles through LDS, one line per thread), same text, same box;
# then the whole default bench line with the new `extras` and the x200 end-to-end row (how long does the driver's run take now?)
set -u
REPO=$PWD
O=$REPO/gpurun_out/r05c5; mkdir -p $O
B="--steps 1 --warmup 0 --matrix 0 --cpu-seconds 0 --e2e 0 --block-sums 0 --scan-carries 0"
for v in main r04pat; do
  if [ "$v" = "main" ]; then LIBENV=""; else LIBENV="WGBSSEG_ALLOW_LIB_OVERRIDE=1 WGBSSEG_LIB=$REPO/tools/micro/_build/libwgbsseg_$v.so"; fi
  env $LIBENV timeout 300 python bench.py $B 2> /dev/null | tail -1 > $O/pat_$v.json
  python -c "
import json; d=json.load(open('$O/pat_$v.json'))['extras']['pat2beta']
print('$v: kernel %.3f ms = %.1f GB/s of text, %.3g reads/s | from host memory %.1f ms = %.0f MB/s | CLI on BGZF %.3f s = %.0f MB/s' % (d['kernel_ms'], d['kernel_text_GB_per_s'], d['kernel_reads_per_s'], d['from_host_memory_wall_s']*1e3, d['from_host_memory_text_MB_per_s'], d['cli_bgzf_wall_s'], d['cli_bgzf_text_MB_per_s']))"
done 2>&1 | tee $O/pat_kernel_ab.txt
SECONDS=0
timeout 900 python bench.py 2> $O/bench_default.err | tail -1 > $O/bench_default.json; echo "default bench: rc $? in $SECONDS s"; python tools/summ.py $O/bench_default.json
python -c "
import json; d=json.load(open('$O/bench_default.json'))
for r in d['matrix']['rows']:
    if r.get('end_to_end'): print('x%d end to end:' % r['samples'], json.dumps(r['end_to_end'])[:600])
print('x32 end to end:', json.dumps(d['end_to_end'])[:400])"
