#!/bin/bash
# Test infrastructure: builds tools/micro/_build/libwgbsseg_carrybug.so = the CURRENT library sources with round 1's carry defect
# (fixed in 8c1419e: a wide scoring tile that wants P[len] alone read one carry past the chunk's) put back, to show that the fuzz
# inside the GPU suite (tests/test_gpu_fuzz.py, test_13b) fails on it:
#     tools/build_carrybug_lib.sh && WGBSSEG_ALLOW_LIB_OVERRIDE=1 WGBSSEG_LIB=$PWD/tools/micro/_build/libwgbsseg_carrybug.so python -m pytest tests/test_gpu_fuzz.py -m gpu -q
set -e
cd "$(dirname "$0")/.."
B=tools/micro/_build/carrybug
rm -rf $B && mkdir -p $B/wgbs_tools_amd $B/include
cp -r wgbs_tools_amd/csrc $B/wgbs_tools_amd/ && cp include/*.h $B/include/ && rm -f $B/wgbs_tools_amd/csrc/*.so
grep -q 'wg_group_start(cd, eA < cd.len ? eA : cd.len - 1)' $B/wgbs_tools_amd/csrc/seg_kernels.h
sed -i 's/wg_group_start(cd, eA < cd.len ? eA : cd.len - 1)/wg_group_start(cd, eA)/' $B/wgbs_tools_amd/csrc/seg_kernels.h
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -pthread $B/wgbs_tools_amd/csrc/wgbsseg.hip -o tools/micro/_build/libwgbsseg_carrybug.so
rm -rf $B
echo built tools/micro/_build/libwgbsseg_carrybug.so
