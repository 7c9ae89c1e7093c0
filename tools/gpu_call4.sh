#!/bin/bash
# round 3, GPU call 4: the recurrence with its SIMD to itself (WGBSSEG_DP_IDLE), the barrier-free multi-process hand-over on the GPU
set -u
O=gpurun_out/c4
mkdir -p $O
B="--matrix 0 --cpu-seconds 0 --e2e 0 --block-sums 0 --steps 10 --warmup 2"
for idle in 0 1 2; do
  WGBSSEG_DP_IDLE=$idle timeout 300 python bench.py $B 2> /dev/null | tail -1 > $O/x32_idle$idle.json
  WGBSSEG_DP_IDLE=$idle timeout 300 python bench.py --samples 8 $B 2> /dev/null | tail -1 > $O/x8_idle$idle.json
  WGBSSEG_DP_IDLE=$idle timeout 300 python bench.py --sites 3527181 $B 2> /dev/null | tail -1 > $O/eighth_idle$idle.json
done
python tools/summ.py $O/x32_idle*.json $O/x8_idle*.json $O/eighth_idle*.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29578 bench.py --gpus 2 --steps 5 --warmup 1 > $O/torchrun2.log 2>&1; echo "torchrun x2 on one GPU: rc $?"; grep '^{' $O/torchrun2.log | tail -1 | cut -c1-330
timeout 300 python -m pytest tests/test_gpu_driver.py -q -m gpu -k "two_ranks" > $O/two_ranks.log 2>&1; echo "two ranks CLI test: rc $? ($(tail -1 $O/two_ranks.log))"
