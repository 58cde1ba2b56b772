#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_net.py tests/test_gpu_baseline_configs.py -m gpu -q --tb=short -p no:cacheprovider -s -k "bf16" > gpurun_out/pytest_p.log 2>&1; grep -E "passed|failed|FAILED|Error|conv1_1 wgrad" gpurun_out/pytest_p.log | cut -c1-200 | tail -10
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --mode parent --precision bf16 --batch 12 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 parent b12:', d['value'], d['sustained']['value'], d['roofline']['families'])"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32x3 b1 (same box):', d['value'], d['sustained']['value'])"
