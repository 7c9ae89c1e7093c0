#!/bin/bash
# round 5, call 4: does overlapping the recurrence with the next stage's scoring pay for SMALL cohorts (x8: recurrence = 16 % of the step)?  WGBSSEG_FORCE_STAGES sweep.
set -u
bash tools/gpu/ab.sh r05c4 "main main@WGBSSEG_FORCE_STAGES=2 main@WGBSSEG_FORCE_STAGES=3 main@WGBSSEG_FORCE_STAGES=4 main@WGBSSEG_FORCE_STAGES=6" "--samples 8;--samples 16;--samples 32" 2>&1 | tee gpurun_out/r05c4_stage_sweep.txt
