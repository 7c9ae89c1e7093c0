#!/bin/bash
# time line of one GPU's share of 8 (and of the whole genome) event by event
set -u
O=gpurun_out/timeline; mkdir -p $O
B="--matrix 0 --cpu-seconds 0 --e2e 0 --extras 0 --block-sums 0 --scan-carries 0 --steps 4 --warmup 2"
WGBSSEG_PROFILE=2 WGBSSEG_PROFILE_STITCH=1 timeout 300 python bench.py $B --sites 3527181 > $O/share.json 2> $O/share.err; python tools/summ.py $O/share.json | cut -c36-; grep "batch of\|stitch\|\[wgbsseg\]" $O/share.err | tail -40
WGBSSEG_PROFILE=2 WGBSSEG_PROFILE_STITCH=1 timeout 300 python bench.py $B > $O/full.json 2> $O/full.err; python tools/summ.py $O/full.json | cut -c36-; grep "batch of\|stitch\|\[wgbsseg\]" $O/full.err | tail -24
