#!/bin/bash
# copy what tools/gpu_final.sh left under gpurun_out/<tag>_final/ into profiles/ (tracked): usage tools/collect_final.sh r05
TAG=${1:-r05}; O=gpurun_out/${TAG}_final; P=profiles
[ -s $O/summary.txt ] || { echo "no $O/summary.txt"; exit 1; }
{ echo "pytest tests -m gpu (FULL tier, no -k, no -x) + __graft_entry__.smoke() + evidence on one MI355X: tools/gpu_final.sh $TAG (the round's last GPU call)"
  cat $O/summary.txt; echo; echo "===== pytest tail"; tail -25 $O/pytest_gpu.log; } > $P/${TAG}_gputest_summary.txt
cp $O/bench_default.json $P/${TAG}_bench_default.json
[ -s $O/bench_driver_cmd.json ] && cp $O/bench_driver_cmd.json $P/${TAG}_bench_driver_cmd.json
cp $O/kernel_stats_configs1.txt $P/${TAG}_bench_fp32x3_kernel_stats.txt
cp $O/kernel_stats_configs2.txt $P/${TAG}_bench_bf16_b12_kernel_stats.txt
cp $O/timeline_configs1.txt $P/${TAG}_step_timeline.txt
cp $O/timeline_configs2.txt $P/${TAG}_step_timeline_bf16_b12.txt
cp $O/profiles/${TAG}_pmc_traffic_*.json $P/ 2>/dev/null
ls -la $P/${TAG}_*
