#!/bin/bash
# the CLI end to end (32 page-cached files -> BED) + the driver tests that compare its BED bytes and stderr with the reference's
set -u
O=gpurun_out/e2e_check; mkdir -p $O
timeout 600 python tools/e2e_bench.py > $O/e2e.log 2>&1; echo "e2e rc $?"; grep "^run\|BED" $O/e2e.log | cut -c1-420
timeout 900 python -m pytest tests/test_gpu_driver.py -q -x -m gpu > $O/driver.log 2>&1; echo "driver tests: rc $? $(tail -1 $O/driver.log)"
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -m gpu -k "hg19_x32" > $O/fullsize32.log 2>&1; echo "fullsize x32: rc $? $(tail -1 $O/fullsize32.log)"
