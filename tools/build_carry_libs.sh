#!/bin/bash
# A/B builds of the library with another carry spacing of k_scan (WG_CARRY_SHIFT: a carry every 2^shift sites):
#     tools/build_carry_libs.sh 8 9 10   ->  tools/micro/_build/libwgbsseg_carry{8,9,10}.so   (use with WGBSSEG_ALLOW_LIB_OVERRIDE=1 WGBSSEG_LIB=...)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/micro/_build
for s in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -pthread -DWG_CARRY_SHIFT=$s wgbs_tools_amd/csrc/wgbsseg.hip -o tools/micro/_build/libwgbsseg_carry$s.so &
done
wait
ls -la tools/micro/_build/libwgbsseg_carry*.so
