#!/bin/bash
# Round 6: the recurrence's roles decoupled by LDS counters instead of one s_barrier per batch (k_dp<7,64,LEAN,FLAGS>, WGBSSEG_DP_FLAGS=1) against the barrier form:
# parity tests under the switch, then bench lines alternating (hg19 x32, x8; islands do not take this kernel).
set -u
O=gpurun_out/dp_flags; mkdir -p $O
# (apply tools/experiments/r06_dp_flags.patch and rebuild first: the switch is not in the product)
WGBSSEG_DP_FLAGS=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "test_05 or test_06 or test_07 or test_08 or test_09 or test_17 or fuzz" > $O/pytest_flags.log 2>&1; echo "parity under WGBSSEG_DP_FLAGS=1: rc $? ($(tail -1 $O/pytest_flags.log))"
B="--matrix 0 --cpu-seconds 0 --e2e 0 --extras 0 --block-sums 0 --scan-carries 0 --steps 10 --warmup 3"
for rep in 1 2 3; do
  for f in 0 1; do
    for a in "" "--samples 8"; do
      tag=$(echo "f${f}_r${rep}_x$a" | tr -d ' -')
      WGBSSEG_DP_FLAGS=$f timeout 300 python bench.py $B $a 2> /dev/null | tail -1 > $O/$tag.json
      echo "flags=$f rep $rep [$a] $(python tools/summ.py $O/$tag.json | cut -c36-)"
    done
  done
done
