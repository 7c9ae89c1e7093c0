#!/bin/bash
# round 6: gated stages for jobs with medium / wide tiles too (two pairs of scoring streams)
set -u
O=gpurun_out/stage_gate; mkdir -p $O
WGBSSEG_STAGE_GATE_SHARED=1 WGBSSEG_FORCE_STAGES=5 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -x -m gpu 2>&1 | tail -4
V="main main@WGBSSEG_STAGE_GATE=0 main@WGBSSEG_LAST_STAGE_PCT=100 main@WGBSSEG_LAST_STAGE_PCT=200 main"
bash tools/gpu/ab.sh stage_gate "$V" "--sites 3527181 --steps 40 --islands;--sites 3527181 --steps 40 --islands --samples 8;--sites 3527181 --steps 20 --islands --samples 100;--sites 3527181 --steps 40" 2>&1 | tee $O/ab10.txt
