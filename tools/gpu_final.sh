#!/bin/bash
# The round's LAST GPU call (a gate, not a habit): the FULL `pytest tests -m gpu` (no -k, no -x) + smoke on the tree as it is,
# then the evidence the docs cite, all from that same tree: per-step PMC traffic of the four bench workloads (copied into profiles/ BEFORE the
# bench runs, so the line carries them), the default bench line, settled rocprofv3 kernel-trace summaries + step timelines of configs[1] / configs[2].
# usage: OSVOS_COMMIT=$(git rev-parse --short HEAD) gpurun --timeout 2400 -- "OSVOS_COMMIT=$OSVOS_COMMIT bash tools/gpu_final.sh r05"
# afterwards: tools/collect_final.sh r05   (copies gpurun_out/<tag>_final/* into profiles/ with the commit hash in the summary)
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp; O=$R/gpurun_out/${TAG}_final; mkdir -p $O
t0=$(date +%s)
echo "tree: ${OSVOS_COMMIT:-unknown}" > $O/summary.txt
python -c "import torch; print(torch.__version__, torch.cuda.get_device_name(0))" >> $O/summary.txt 2>&1
timeout 2100 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=15 > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/summary.txt
tail -3 $O/pytest_gpu.log >> $O/summary.txt
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/summary.txt; tail -1 $O/smoke.log >> $O/summary.txt
echo "tier+smoke $(( $(date +%s) - t0 )) s" >> $O/summary.txt
pmc() { # key label args...
  K=$1; L=$2; shift; shift
  bash tools/gpu_pmc_step.sh ${TAG}_final/pmc_$K "$L" --no-parity "$@" > /dev/null 2>&1
  [ -s $O/pmc_$K/traffic.json ] && cp $O/pmc_$K/traffic.json profiles/${TAG}_pmc_traffic_$K.json
  echo "pmc $K $(( $(date +%s) - t0 )) s" >> $O/summary.txt
}
pmc configs1 "configs[1] 854x480 b1 online f32x3"
pmc configs2 "configs[2] 854x480 b12 parent bf16" --mode parent --precision bf16 --batch 12
pmc configs4 "configs[4] 1920x1080 b4 forward f32x3 (eager launches of the graph's kernels)" --mode infer --height 1080 --width 1920 --batch 4 --graph 0
pmc window_fused "configs[1] window-fused (5 frames per step) f32x3" --window-fused 1
mkdir -p $O/profiles; cp profiles/${TAG}_pmc_traffic_*.json $O/profiles/ 2>/dev/null
# the line the driver reads (its exact command) FIRST; bench.py prints the compact line and leaves the full record in gpurun_out/bench_detail.json
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench (driver's command: --gpus 1 --steps 20 --warmup 5) exit $? $(( $(date +%s) - t0 )) s, line $(wc -c < $O/bench_driver_cmd.json) bytes" >> $O/summary.txt
cp gpurun_out/bench_detail.json $O/bench_detail_driver_cmd.json 2>/dev/null
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench exit $? $(( $(date +%s) - t0 )) s" >> $O/summary.txt
cp gpurun_out/bench_detail.json $O/bench_detail_default.json 2>/dev/null
NOBENCH=1 bash tools/gpu_evidence.sh ${TAG}_final > /dev/null 2>&1
# round 6 extras: per-layer tables, the Cin = 64 tile sweep, per-tensor gradient errors of the precisions, the seam of the two-piece backward
bash tools/gpu_run.sh ${TAG}_final "layers:bf16 12 --all" "layers:x3 1 --all" > $O/layers.log 2>&1
timeout 120 python tools/tune_p64.py --tiles 9,109,36,136,38,138 > $O/tune_p64.txt 2>&1
timeout 300 python tools/grad_error_table.py 2>&1 | grep -v "amdgpu.ids\|Constructing\|Initializing" > $O/grad_error_table.txt
# FP16-pair precisions: the f16 pipe next to the bf16 one, per-layer ReLU flips of every fp32-family precision against float64
timeout 120 tools/native/bin/mfma_f16_probe > $O/mfma_f16_probe.txt 2>&1
{ timeout 300 python tools/net_flip_probe.py 240 427; timeout 300 python tools/net_flip_probe.py 480 854; } 2>&1 | grep -v "amdgpu.ids\|Constructing\|Initializing" > $O/net_flips.txt
echo "extras $(( $(date +%s) - t0 )) s" >> $O/summary.txt
echo "total $(( $(date +%s) - t0 )) s" >> $O/summary.txt
cat $O/summary.txt
python - <<PY
import json
d = json.loads([l for l in open("$O/bench_default.json") if l.startswith("{")][-1])
print("headline", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "traffic", (d["roofline"].get("traffic") or {}).get("conv_family"), "parity ok", (d.get("parity") or {}).get("within_bars"))
for e in d.get("extra_configs") or []:
    r = e.get("roofline") or {}
    print(e.get("config", "")[:44], e.get("value"), "frac", r.get("frac"), "traffic", ((r.get("traffic") or {}).get("conv_family") or {}).get("ratio"), "parity", (e.get("parity") or {}).get("within_bars"))
PY
