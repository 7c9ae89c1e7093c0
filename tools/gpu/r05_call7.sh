#!/bin/bash
# round 5, call 7: upload threads pinned to the CPUs next to the GPU (WGBSSEG_UPLOAD_PIN, default on) against free-running threads, x200 and x32 end to end
set -u
O=$PWD/gpurun_out/r05c7; mkdir -p $O
python - <<'PY' > $O/topology.txt 2>&1
import glob, os, torch
print('cpus allowed:', len(os.sched_getaffinity(0)), 'of', os.cpu_count())
for d in sorted(glob.glob('/sys/devices/system/node/node*')):
    print(os.path.basename(d), open(d + '/cpulist').read().strip())
import ctypes
hip = ctypes.CDLL('libamdhip64.so')
buf = ctypes.create_string_buffer(64); hip.hipDeviceGetPCIBusId(buf, 64, 0); bus = buf.value.decode().lower()
print('device 0 bus id', bus)
for f in ('local_cpulist', 'numa_node'):
    try: print(f, open('/sys/bus/pci/devices/%s/%s' % (bus, f)).read().strip())
    except Exception as e: print(f, 'unreadable:', e)
PY
cat $O/topology.txt
run() { n=$1; shift; env "$@" timeout 600 python tools/e2e_bench.py --samples $SAMPLES --keep > $O/e2e_x${SAMPLES}_$n.log 2>&1; echo "x$SAMPLES $n: rc $?"; grep "^run\|streaming" $O/e2e_x${SAMPLES}_$n.log | sed 's/\[wt segment\] found [0-9,]* blocks | //' | cut -c1-330; }
SAMPLES=200
run pinned WGBSSEG_NOP=1
run free WGBSSEG_UPLOAD_PIN=0
run pinned_t8_4MB WGBSSEG_UPLOAD_THREADS=8 WGBSSEG_UPLOAD_PIECE_KB=4096
run free_t8_4MB WGBSSEG_UPLOAD_PIN=0 WGBSSEG_UPLOAD_THREADS=8 WGBSSEG_UPLOAD_PIECE_KB=4096
run pinned_t8 WGBSSEG_UPLOAD_THREADS=8
run pinned_t6_2MB WGBSSEG_UPLOAD_THREADS=6 WGBSSEG_UPLOAD_PIECE_KB=2048
rm -rf /tmp/wgbs_e2e
SAMPLES=32
run pinned WGBSSEG_NOP=1
run free WGBSSEG_UPLOAD_PIN=0
run pinned_t8_4MB WGBSSEG_UPLOAD_THREADS=8 WGBSSEG_UPLOAD_PIECE_KB=4096
rm -rf /tmp/wgbs_e2e
