#!/bin/bash
# round 6: the stages of a share alternating between the two scoring streams behind k_stage_gate, and a last stage of another length — the defaults
set -u
O=gpurun_out/stage_gate; mkdir -p $O
bash tools/gpu/check.sh stage_gate_check 2>&1 | tee $O/check.txt
V="main main@WGBSSEG_STAGE_GATE=0 main"
bash tools/gpu/ab.sh stage_gate "$V" "--sites 3527181 --steps 40;--sites 3527181 --steps 40 --samples 8;--sites 3527181 --steps 40 --samples 16;--sites 3527181 --steps 40 --samples 24;--sites 3527181 --steps 20 --samples 64;--sites 3527181 --steps 10 --samples 200;--sites 3527181 --steps 40 --islands;--sites 7000000 --steps 20" 2>&1 | tee $O/ab8.txt
