#!/bin/bash
# A/B of library builds (tools/build_variants.sh) on one box:  bash tools/gpu/ab.sh OUTDIR "variant ..." "bench-arg-set;bench-arg-set;..."
# every variant x every argument set: bench.py --steps 10 --warmup 3, device time per kernel in one line each (tools/summ.py)
set -u
REPO=$PWD
O=$REPO/gpurun_out/$1; mkdir -p $O
B="--matrix 0 --cpu-seconds 0 --e2e 0 --block-sums 0 --steps 10 --warmup 3"
IFS=';' read -ra SETS <<< "$3"
for v in $2; do
  if [ "$v" = "main" ]; then unset WGBSSEG_LIB; else export WGBSSEG_LIB=$REPO/tools/micro/_build/libwgbsseg_$v.so; fi
  i=0
  for a in "${SETS[@]}"; do
    f=$O/${v}_$i.json
    timeout 300 python bench.py $B $a 2> $O/${v}_$i.err | tail -1 > $f
    echo "== $v [$a]"; python tools/summ.py $f 2>&1 | cut -c36-
    i=$((i+1))
  done
done
