#!/bin/bash
# round 6: BASELINE.json configs[4] at full size (28 M x 512, max_cpg 5000: 157 stages) with the gated stages and without
set -u
O=$PWD/gpurun_out/deep_gate; mkdir -p $O
for g in 768 0; do
  WGBSSEG_STAGE_GATE=$g WGBSSEG_PROFILE=2 WGBSSEG_DEEP_ORACLE_CHUNKS=1 timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -x -m gpu -k deep_full_genome -s > $O/deep_gate_$g.log 2>&1
  echo "gate $g: rc $? $(tail -1 $O/deep_gate_$g.log)"; grep -m2 "stages" $O/deep_gate_$g.log | cut -c1-200; grep "deep full genome" $O/deep_gate_$g.log | cut -c1-400
  cp gpurun_out/deep_full_timing.json $O/deep_full_timing_gate_$g.json 2>/dev/null
done
