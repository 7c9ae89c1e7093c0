#!/bin/bash
# The round's bench lines alone (the PMC / instruction-mix files of this source state are under profiles/ already): 20 steps, then the driver's flags.  -> gpurun_out/lines/
set -u
O=gpurun_out/lines; mkdir -p $O
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
timeout 900 python bench.py --steps 20 --warmup 5 2> $O/bench.err | tail -1 > $O/bench.json; python tools/summ.py $O/bench.json | head -1
SECONDS=0; timeout 900 python bench.py 2> $O/bench_default_flags.err | tail -1 > $O/bench_default_flags.json; echo "driver's flags: $SECONDS s"; python tools/summ.py $O/bench_default_flags.json | head -1
