#!/bin/bash
# every job of the parity / fuzz tests in five GATED stages (WGBSSEG_STAGE_GATE_SHARED=1: whatever the tiles, whoever else lives on the device), then the aligned fuzz the same way
set -u
O=gpurun_out/gated_forced; mkdir -p $O
export WGBSSEG_STAGE_GATE_SHARED=1 WGBSSEG_FORCE_STAGES=5
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -x -m gpu > $O/pytest.log 2>&1; echo "parity + fuzz tests, five gated stages: rc $? ($(tail -1 $O/pytest.log))"
FS=$(python -c "import sys; sys.path.insert(0, 'tests'); import fuzzlib; print(fuzzlib.round_number() * 1000000 + 700000)")
timeout 200 python tools/aligned_fuzz.py $FS 1000000 120 > $O/fuzz_aligned.log 2>&1; echo "aligned fuzz from seed $FS, five gated stages: $(tail -1 $O/fuzz_aligned.log)"
WGBSSEG_FORCE_STAGES=3 WGBSSEG_LAST_STAGE_PCT=300 timeout 200 python tools/extra_fuzz.py $((FS / 100)) 100000 60 > $O/fuzz_uniform.log 2>&1; echo "uniform fuzz, three gated stages with a long last one: $(tail -1 $O/fuzz_uniform.log)"
