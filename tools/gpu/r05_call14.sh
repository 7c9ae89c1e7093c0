#!/bin/bash
# round 5, call 14: four recurrence stages for SMALL cohorts (call 4 saw x8 10.61 -> 10.13 ms): repeated, alternating, on one box, x4 / x8 / x12 / x16
set -u
mkdir -p gpurun_out/r05c14
for rep in 1 2 3; do
  bash tools/gpu/ab.sh r05c14 "main main@WGBSSEG_FORCE_STAGES=4" "--samples 8"
done 2>&1 | tee gpurun_out/r05c14_small_cohort_stages.txt
bash tools/gpu/ab.sh r05c14 "main main@WGBSSEG_FORCE_STAGES=4 main main@WGBSSEG_FORCE_STAGES=4" "--samples 4;--samples 12;--samples 16" 2>&1 | tee -a gpurun_out/r05c14_small_cohort_stages.txt
