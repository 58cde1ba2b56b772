#!/bin/bash
# First GPU pass over the FP16-pair ('fp32h2') kernels: pipe probe, op-level tests next to the exact fp32 kernels, network tests, bench rows.
cd "$(dirname "$0")/.."
O=gpurun_out/h2; mkdir -p $O
timeout 120 tools/native/bin/mfma_f16_probe > $O/probe.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "fp16_pairs or two_pieces" -s > $O/ops.txt 2>&1; tail -3 $O/ops.txt
timeout 900 python -m pytest tests/test_gpu_net.py -q -k "h2" > $O/net.txt 2>&1; tail -3 $O/net.txt
for p in fp32x3 fp32x3b2 fp32h2 fp32x3h2; do
  timeout 300 python bench.py --precision $p --no-extra --no-cpu-baseline --steps 30 --warmup 5 --full-line > $O/bench_$p.txt 2>&1
  python - <<PY
import json
try:
    l = json.loads(open("$O/bench_$p.txt").read().strip().splitlines()[-1])
    print("$p", l["value"], l["ms_per_step"], json.dumps(l.get("parity"))[:600])
except Exception as e:
    print("$p failed", e)
PY
done
for p in fp32x3 fp32h2; do
  timeout 300 python bench.py --mode infer --height 1080 --width 1920 --batch 4 --graph 1 --precision $p --no-extra --no-cpu-baseline --steps 20 --warmup 5 --full-line > $O/bench4_$p.txt 2>&1
  python - <<PY
import json
try:
    l = json.loads(open("$O/bench4_$p.txt").read().strip().splitlines()[-1])
    print("configs[4] $p", l["value"], l["ms_per_step"], json.dumps(l.get("parity"))[:400])
except Exception as e:
    print("configs[4] $p failed", e)
PY
done
timeout 900 python -m pytest tests/test_gpu_trained_like.py tests/test_gpu_baseline_configs.py -q -k "h2" > $O/more.txt 2>&1; tail -3 $O/more.txt
