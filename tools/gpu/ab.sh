#!/bin/bash
# A/B of library builds (tools/build_variants.sh) and environment switches on one box:
#   bash tools/gpu/ab.sh OUTDIR "variant[@ENV=val[,ENV=val]] ..." "bench-arg-set;bench-arg-set;..."
# variant = main (csrc/libwgbsseg.so) or a name under tools/micro/_build/libwgbsseg_<name>.so; every variant x every argument set:
# bench.py --steps 10 --warmup 3, device time per kernel in one line each (tools/summ.py)
set -u
REPO=$PWD
O=$REPO/gpurun_out/$1; mkdir -p $O
B="--matrix 0 --cpu-seconds 0 --e2e 0 --extras 0 --block-sums 0 --scan-carries 0 --steps 10 --warmup 3"
IFS=';' read -ra SETS <<< "$3"
for spec in $2; do
  v=${spec%%@*}; envs=""; [ "$spec" != "$v" ] && envs=${spec#*@}
  if [ "$v" = "main" ]; then LIBENV=""; else LIBENV="WGBSSEG_ALLOW_LIB_OVERRIDE=1 WGBSSEG_LIB=$REPO/tools/micro/_build/libwgbsseg_$v.so"; fi
  i=0
  for a in "${SETS[@]}"; do
    tag=$(echo "${spec}_$i" | tr '@=,/' '____')
    f=$O/$tag.json
    env $LIBENV $(echo $envs | tr ',' ' ') timeout 300 python bench.py $B $a 2> $O/$tag.err | tail -1 > $f
    echo "== $spec [$a]"; python tools/summ.py $f 2>&1 | cut -c36-
    i=$((i+1))
  done
done
