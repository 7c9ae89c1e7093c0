#!/bin/bash
# round 5, call 8: persistent scoring workgroups (WGBSSEG_COST_PERSIST=1: as many workgroups as fit the chip, each walking tiles of its XCD's groups off a counter)
# against one tile per workgroup, and against the library before the kernel was given its tile loop ("prev"); parity of the persistent form
set -u
O=$PWD/gpurun_out/r05c8; mkdir -p $O
WGBSSEG_COST_PERSIST=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "test_05 or test_06 or test_08 or test_09 or test_16 or test_17 or test_07" > $O/tests_persist.log 2>&1; echo "parity (persistent): rc $? ($(tail -1 $O/tests_persist.log))"
bash tools/gpu/ab.sh r05c8 "prev main main@WGBSSEG_COST_PERSIST=1" "--samples 8;--samples 32;--sites 3527181;--samples 32 --islands" 2>&1 | tee $O/persist_ab.txt
