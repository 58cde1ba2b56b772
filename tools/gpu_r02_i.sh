#!/bin/bash
# Round-2 call I: full GPU tier, bf16 weight-gradient form A/B at the configs[2] workload, final default bench line.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_gpu.log | tail -12
for f in 2 3; do
OSVOS_WGRAD_FORM=$f timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --mode parent --precision bf16 --batch 12 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 parent b12 wgrad form $f:', d['value'], d['sustained'], d['roofline']['families'])"
done
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | cut -c1-300
