#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py -m gpu -q --tb=short -p no:cacheprovider -k "wgrad or golden or sgd or partially or forms" > gpurun_out/pytest_q.log 2>&1; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_q.log | tail -6
for t in 1 0; do
OSVOS_WGRAD_REDUCE_T=$t timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --mode parent --precision bf16 --batch 12 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 parent b12 reduce_t=$t:', d['value'], d['sustained']['value'], d['roofline']['families'])"
OSVOS_WGRAD_REDUCE_T=$t timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32x3 b1 reduce_t=$t:', d['value'], d['sustained']['value'], d['roofline']['families'])"
done
