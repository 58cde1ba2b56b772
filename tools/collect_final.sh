#!/bin/bash
# copy what tools/gpu_final.sh left under gpurun_out/<tag>_final/ into profiles/ (tracked): usage tools/collect_final.sh r05
TAG=${1:-r06}; O=gpurun_out/${TAG}_final; P=profiles
[ -s $O/summary.txt ] || { echo "no $O/summary.txt"; exit 1; }
{ echo "pytest tests -m gpu (FULL tier, no -k, no -x) + __graft_entry__.smoke() + evidence on one MI355X: tools/gpu_final.sh $TAG (the round's last GPU call)"
  cat $O/summary.txt; echo; echo "===== pytest tail"; tail -25 $O/pytest_gpu.log; } > $P/${TAG}_gputest_summary.txt
# the line the driver's command printed is THE round's line (tests/test_bench_contract_cpu.py reads it); the full records ride along
cp $O/bench_driver_cmd.json $P/${TAG}_bench_default.json
[ -s $O/bench_detail_driver_cmd.json ] && cp $O/bench_detail_driver_cmd.json $P/${TAG}_bench_detail.json
[ -s $O/bench_default.json ] && cp $O/bench_default.json $P/${TAG}_bench_50steps.json
for f in layer_table_bf16 layer_table_x3; do [ -s $O/$f.txt ] && cp $O/$f.txt $P/${TAG}_${f}_all.txt; done
[ -s $O/tune_p64.txt ] && cp $O/tune_p64.txt $P/${TAG}_tune_p64.txt
[ -s $O/grad_error_table.txt ] && cp $O/grad_error_table.txt $P/${TAG}_grad_error_table.txt
cp $O/kernel_stats_configs1.txt $P/${TAG}_bench_fp32x3_kernel_stats.txt
cp $O/kernel_stats_configs2.txt $P/${TAG}_bench_bf16_b12_kernel_stats.txt
cp $O/timeline_configs1.txt $P/${TAG}_step_timeline.txt
cp $O/timeline_configs2.txt $P/${TAG}_step_timeline_bf16_b12.txt
[ -s $O/kernel_stats_configs1_fp32x3b2.txt ] && cp $O/kernel_stats_configs1_fp32x3b2.txt $P/${TAG}_bench_fp32x3b2_kernel_stats.txt
[ -s $O/timeline_configs1_fp32x3b2.txt ] && cp $O/timeline_configs1_fp32x3b2.txt $P/${TAG}_step_timeline_fp32x3b2.txt
[ -s $O/kernel_stats_configs1_fp32h2.txt ] && cp $O/kernel_stats_configs1_fp32h2.txt $P/${TAG}_bench_fp32h2_kernel_stats.txt
[ -s $O/mfma_f16_probe.txt ] && cp $O/mfma_f16_probe.txt $P/${TAG}_mfma_f16_probe.txt
[ -s $O/net_flips.txt ] && cp $O/net_flips.txt $P/${TAG}_net_flips.txt
cp $O/profiles/${TAG}_pmc_traffic_*.json $P/ 2>/dev/null
ls -la $P/${TAG}_*
