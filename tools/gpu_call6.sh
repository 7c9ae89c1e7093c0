#!/bin/bash
set -u
O=gpurun_out/c6; mkdir -p $O
timeout 300 python tools/dp_timing.py 480 8 2>&1 | grep -v amdgpu.ids | tee $O/dp_timing_480.txt
timeout 300 python tools/dp_timing.py 240 8 2>&1 | grep -v amdgpu.ids | tee $O/dp_timing_240.txt
