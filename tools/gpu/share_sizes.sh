#!/bin/bash
# one GPU's share of a run on 2 / 4 / 8 GPUs (a half, a quarter, an eighth of the hg19-shaped genome) at x 32 and x 200, beside the whole genome: what one GPU can say about the 1 -> 8 curve
set -u
O=gpurun_out/share_sizes; mkdir -p $O
bash tools/gpu/ab.sh share_sizes "main" "--steps 10;--sites 14108724 --steps 20;--sites 7054362 --steps 20;--sites 3527181 --steps 40;--samples 200 --steps 3;--samples 200 --sites 14108724 --steps 5;--samples 200 --sites 7054362 --steps 10;--samples 200 --sites 3527181 --steps 10" 2>&1 | tee $O/sizes.txt
