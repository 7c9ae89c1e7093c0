#!/bin/bash
# round 3, GPU call 3: medium tiles after the staging rewrite (parity, sample-group sweep), the multi-process bench mode on the GPU
set -u
O=gpurun_out/c3
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "test_16 or test_05 or test_06 or test_07 or test_13" > $O/parity_subset.log 2>&1; echo "parity subset: rc $? ($(tail -1 $O/parity_subset.log))"
WGBSSEG_FUZZ_SECONDS=30 timeout 300 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -s -k "time_boxed" > $O/fuzz.log 2>&1; echo "fuzz: rc $? ($(tail -1 $O/fuzz.log)) $(grep 'aligned fuzz' $O/fuzz.log)"
B="--matrix 0 --cpu-seconds 0 --e2e 0 --block-sums 0 --steps 8 --warmup 2"
timeout 300 python bench.py --islands $B 2> $O/isl.err | tail -1 > $O/isl_default.json
for ns in 4 8 16 32; do WGBSSEG_NSM=$ns timeout 300 python bench.py --islands $B 2> /dev/null | tail -1 > $O/isl_nsm$ns.json; done
timeout 300 python bench.py --islands --samples 8 $B 2> /dev/null | tail -1 > $O/isl_x8.json
WGBSSEG_MEDIUM_WMAX=0 timeout 300 python bench.py --islands --samples 8 $B 2> /dev/null | tail -1 > $O/isl_x8_nomedium.json
python tools/summ.py $O/isl_default.json $O/isl_nsm4.json $O/isl_nsm8.json $O/isl_nsm16.json $O/isl_nsm32.json $O/isl_x8.json $O/isl_x8_nomedium.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --steps 5 --warmup 1 > $O/torchrun2.log 2>&1; echo "torchrun x2 on one GPU: rc $?"; grep '^{' $O/torchrun2.log | tail -1 | cut -c1-700
timeout 600 python bench.py --gpus 2 --steps 5 --warmup 1 --matrix 0 --cpu-seconds 0 --e2e 0 > $O/group2.log 2>&1; echo "share group x2 on one GPU: rc $?"; grep '^{' $O/group2.log | tail -1 | cut -c1-400
