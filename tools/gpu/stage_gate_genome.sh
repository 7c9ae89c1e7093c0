#!/bin/bash
# round 6: does the gate change the verdict on staging the WHOLE genome (483 chunks; one stage since round 2)?
set -u
O=gpurun_out/stage_gate; mkdir -p $O
V="main"
for st in 2 4 8; do for p in 100 300; do V="$V main@WGBSSEG_FORCE_STAGES=$st,WGBSSEG_LAST_STAGE_PCT=$p"; done; done
V="$V main@WGBSSEG_FORCE_STAGES=4,WGBSSEG_STAGE_GATE=0 main"
bash tools/gpu/ab.sh stage_gate_genome "$V" "--steps 10;--steps 20 --samples 8" 2>&1 | tee $O/ab9.txt
