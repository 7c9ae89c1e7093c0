#!/bin/bash
# Round 6: the upload of the CLI end to end (page-cached .beta files -> HBM) under the variants of csrc/wgbsseg.hip::fill_piece / upload_rows_streaming:
#   pages of a piece mapped in one call (POPULATE), non-temporal fill of the staging piece (NT), staging pieces per thread (DEPTH), threads, piece size.
#   bash tools/gpu/upload_ab.sh [samples ...]     -> gpurun_out/upload_ab/e2e_x<samples>_<variant>.log, one line per run on stdout
set -u
O=gpurun_out/upload_ab; mkdir -p $O
numactl -H 2>/dev/null | head -6; lscpu | grep -i "numa\|model name\|socket" | head -8
for N in ${@:-32 200}; do
  for spec in "default:X=0" "populate:WGBSSEG_UPLOAD_POPULATE=1" "nt:WGBSSEG_UPLOAD_NT=1" "populate_nt:WGBSSEG_UPLOAD_POPULATE=1 WGBSSEG_UPLOAD_NT=1" \
              "depth4:WGBSSEG_UPLOAD_DEPTH=4" "populate_nt_depth3:WGBSSEG_UPLOAD_POPULATE=1 WGBSSEG_UPLOAD_NT=1 WGBSSEG_UPLOAD_DEPTH=3" \
              "t8_2MB:WGBSSEG_UPLOAD_THREADS=8 WGBSSEG_UPLOAD_PIECE_KB=2048" "t8_2MB_populate_nt:WGBSSEG_UPLOAD_THREADS=8 WGBSSEG_UPLOAD_PIECE_KB=2048 WGBSSEG_UPLOAD_POPULATE=1 WGBSSEG_UPLOAD_NT=1" \
              "t12_2MB_populate_nt:WGBSSEG_UPLOAD_THREADS=12 WGBSSEG_UPLOAD_PIECE_KB=2048 WGBSSEG_UPLOAD_POPULATE=1 WGBSSEG_UPLOAD_NT=1" \
              "t16_1MB_populate_nt:WGBSSEG_UPLOAD_THREADS=16 WGBSSEG_UPLOAD_PIECE_KB=1024 WGBSSEG_UPLOAD_POPULATE=1 WGBSSEG_UPLOAD_NT=1" "nopin:WGBSSEG_UPLOAD_PIN=0"; do
    n=${spec%%:*}; e=${spec#*:}
    env $e timeout 900 python tools/e2e_bench.py --samples $N --keep > $O/e2e_x${N}_$n.log 2>&1
    echo "x$N $n: rc $? | $(grep '^run [12]' $O/e2e_x${N}_$n.log | sed 's/rc 0, //; s/CpG-sites.s end to end; //' | cut -c1-210 | tr '\n' '|')"
  done
  grep BED $O/e2e_x${N}_default.log
  rm -rf /tmp/wgbs_e2e
done
