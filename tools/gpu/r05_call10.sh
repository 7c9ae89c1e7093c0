#!/bin/bash
# round 5, call 10: the block reduction with the .bin rescale in integers (231 -> 211 VALU per tile and wavefront, no fp64): parity, then the bench's block_sums entry
set -u
O=$PWD/gpurun_out/r05c10; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_markers.py -q -x > $O/tests_blocks.log 2>&1; echo "block tests: rc $? ($(tail -1 $O/tests_blocks.log))"
for i in 1 2; do
timeout 600 python bench.py --steps 5 --warmup 2 --matrix 0 --cpu-seconds 0 --e2e 0 --extras 0 --scan-carries 0 2> /dev/null | tail -1 > $O/bench_bs_$i.json
python -c "
import json; d=json.load(open('$O/bench_bs_$i.json'))['block_sums']; print('block_sums: .bin rows %.4f ms (all %s) = %.0f GB/s = %.3f of peak; means %.4f, raw %.4f' % (d['ms_bin_rows'], ['%.3f' % x for x in d['ms_bin_rows_all']], d['GB_per_s'], d['frac_of_hbm_peak'], d['ms_means'], d['ms_raw_sums']))"
done
