#!/bin/bash
# Final measurements, part B: both N > 1 forms on one GPU, the whole genome x200 (and the adversarial world x8 / x32) against the reference binary on the final library,
# a fuzz on this round's seeds.  gpurun_out/final/.
set -u
REPO=$PWD
O=$REPO/gpurun_out/final
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_integration_stub.py -q > $O/tests_stub.log 2>&1; echo "integration stub (as printed in INTEGRATION.md): rc $? ($(tail -1 $O/tests_stub.log))"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29579 bench.py --gpus 2 --oversubscribe --steps 5 --warmup 1 > $O/torchrun2.log 2>&1; echo "torchrun x2: rc $?"
WGBSSEG_BENCH_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29580 bench.py --gpus 1 --steps 5 --warmup 1 --matrix 0 > $O/torchrun1_rccl.log 2>&1; echo "torchrun x1 (RCCL group): rc $?"
timeout 600 python bench.py --gpus 8 --oversubscribe --steps 5 --warmup 1 > $O/group8_on_one_gpu.log 2>&1; echo "group of 8 shares on one GPU: rc $? $(tail -1 $O/group8_on_one_gpu.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('n_gpus', d['n_gpus'], 'shares', d['config']['shares'], d['ms_per_step'], d['config']['share_work_max_over_mean'], (d.get('matrix') or {}).get('rows'))")"
tail -1 $O/torchrun2.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('torchrun x2: n_gpus', d['n_gpus'], 'shares', d['config']['shares'], 'oversubscribed', d['config']['oversubscribed'], '%.2f ms' % d['ms_per_step'])"
timeout 1500 python tools/full_vs_reference.py --samples 200 > $O/x200_full_vs_reference.log 2>&1; echo "full x200 vs reference: rc $? $(tail -1 $O/x200_full_vs_reference.log | cut -c1-330)"
for n in 8 32; do timeout 900 python tools/full_vs_reference.py --samples $n --adversarial > $O/adversarial_x${n}_full_vs_reference.log 2>&1; echo "adversarial x$n: rc $? $(tail -1 $O/adversarial_x${n}_full_vs_reference.log | cut -c1-330)"; done
FS=$(python -c "import sys; sys.path.insert(0, 'tests'); import fuzzlib; print(fuzzlib.round_number() * 1000000 + 400000)")
timeout 200 python tools/aligned_fuzz.py $FS 1000000 150 > $O/fuzz_long_aligned.log 2>&1; echo "aligned fuzz from seed $FS: $(tail -1 $O/fuzz_long_aligned.log)"
timeout 120 python tools/extra_fuzz.py $((FS / 100)) 100000 80 > $O/fuzz_long_uniform.log 2>&1; echo "uniform fuzz: $(tail -1 $O/fuzz_long_uniform.log)"
timeout 120 python tools/disorder_fuzz.py $((FS / 100)) 100000 60 > $O/fuzz_disorder.log 2>&1; echo "disorder fuzz: $(tail -1 $O/fuzz_disorder.log)"
