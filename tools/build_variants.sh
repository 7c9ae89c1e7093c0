#!/bin/bash
# A/B builds of libwgbsseg.so into tools/micro/_build/ (they travel with gpurun; WGBSSEG_ALLOW_LIB_OVERRIDE=1 WGBSSEG_LIB=... selects one):
#   tools/build_variants.sh name1 "-DFLAG=1 ..." name2 "..." ...
set -e
cd "$(dirname "$0")/../wgbs_tools_amd/csrc"
mkdir -p ../../tools/micro/_build
while [ $# -ge 2 ]; do
  n=$1; f=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -pthread $f wgbsseg.hip -o ../../tools/micro/_build/libwgbsseg_$n.so 2>/dev/null &
done
wait
ls -la ../../tools/micro/_build/*.so
