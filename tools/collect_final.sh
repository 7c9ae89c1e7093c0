#!/bin/bash
# gpurun_out/final/ (written by tools/gpu/final_a.sh on the GPU box) -> profiles/rNN_*   Usage: tools/collect_final.sh [r04]
R=${1:-r06}
cd "$(dirname "$0")/.."
O=gpurun_out/final
for f in cost_traffic scan_traffic scan_traffic_carries scan_traffic_islands pmc_cost_sq; do cp $O/$f.json profiles/${R}_$f.json; done
cp $O/bench.json profiles/${R}_bench_28M_x32.json
cp $O/bench_islands.json profiles/${R}_bench_28M_x32_islands.json
cp $O/bench_islands_x8.json profiles/${R}_bench_28M_x8_islands.json
cp $O/bench_one_eighth.json profiles/${R}_bench_one_eighth.json
cp $O/rocprofv3_kernel_stats_28M_x32.csv profiles/${R}_rocprofv3_kernel_stats_28M_x32.csv
cp $O/rocprofv3_kernel_stats_28M_x32_islands.csv profiles/${R}_rocprofv3_kernel_stats_28M_x32_islands.csv
cp $O/torchrun2.log profiles/${R}_torchrun2_on_one_gpu.log
cp $O/torchrun1_rccl.log profiles/${R}_torchrun1_rccl.log
cp $O/group8_on_one_gpu.log profiles/${R}_group8_on_one_gpu.log
cp $O/gpu_tests_all.log profiles/${R}_gpu_tests_all.log
[ -f $O/fuzz_long_aligned.log ] && cat $O/fuzz_long_aligned.log $O/fuzz_long_uniform.log > profiles/${R}_fuzz_long.log
python - <<PY
import json
d = json.load(open('profiles/${R}_bench_28M_x32.json'))
print(d['config']['csrc_sha'], '%.2f ms/step' % d['ms_per_step'], 'k_cost %.2f ms frac %.3f traffic %s' % (d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['traffic']),
      'scan %.2f' % d['roofline_scan']['frac'], 'carries %.2f x%.3f' % (d['roofline_scan_carries']['frac'], d['roofline_scan_carries']['traffic_over_algorithmic'] or 0), 'e2e %.3f s' % d['end_to_end']['wall_s'])
PY
tail -1 profiles/${R}_gpu_tests_all.log
