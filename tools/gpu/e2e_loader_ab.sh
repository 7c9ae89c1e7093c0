#!/bin/bash
# end to end through the CLI — the upload from mapped rows (rounds 2-3) against pread into page-locked staging, 4 / 8 / 16 threads
set -u
O=gpurun_out/e2e_loader_ab; mkdir -p $O
timeout 600 python tools/e2e_bench.py --keep > $O/e2e_pread8.log 2>&1; echo "pread 8 (default): rc $?"; grep "^run\|BED" $O/e2e_pread8.log | cut -c1-400
for spec in "mapped:WGBSSEG_UPLOAD_MAPPED=1" "pread4:WGBSSEG_UPLOAD_THREADS=4" "pread16:WGBSSEG_UPLOAD_THREADS=16" "pread8_2MB:WGBSSEG_UPLOAD_PIECE_KB=2048" "pread12_512KB:WGBSSEG_UPLOAD_THREADS=12 WGBSSEG_UPLOAD_PIECE_KB=512"; do
  n=${spec%%:*}; e=${spec#*:}
  env $e timeout 600 python tools/e2e_bench.py --keep > $O/e2e_$n.log 2>&1; echo "$n: rc $?"; grep "^run\|BED" $O/e2e_$n.log | cut -c1-400
done
rm -rf /tmp/wgbs_e2e
