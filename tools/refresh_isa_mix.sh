#!/bin/bash
# Re-derive the instruction mix of the scoring kernel's evaluation from the ISA of the CURRENT sources (no GPU needed) into
# profiles/rNN_cost_isa_mix_{narrow,narrow128,wide}.json — bench.py reports `roofline.issue` only from files whose csrc_sha is
# the library's, so run this after every change under wgbs_tools_amd/csrc/ or include/.   Usage: tools/refresh_isa_mix.sh [r05]
R=${1:-r05}
cd "$(dirname "$0")/.."
python tools/micro/count_cost_loop.py 64 3 0 --json profiles/${R}_cost_isa_mix_narrow.json | tail -1
python tools/micro/count_cost_loop.py 128 3 0 --json profiles/${R}_cost_isa_mix_narrow128.json | tail -1
python tools/micro/count_cost_loop.py 16 2 1 --json profiles/${R}_cost_isa_mix_wide.json | tail -1
python tools/micro/count_cost_loop.py 16 3 2 --json profiles/${R}_cost_isa_mix_medium.json | tail -1
