#!/bin/bash
# round 4, call 3: the plain path for non-ascending loci + the one-group form of k_cost: parity file
set -u
O=gpurun_out/r04c3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "test_19 or test_11" > $O/t19.log 2>&1; echo "test_19/11 rc $? $(tail -1 $O/t19.log)"
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x > $O/parity.log 2>&1; echo "parity rc $? $(tail -1 $O/parity.log)"
tail -30 $O/t19.log
