#!/bin/bash
# Round-2 call H: pre-split f32x3 weights, head op tests, generic head; sweep; bench.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py -m gpu -q --tb=short -p no:cacheprovider -s -k "f32x3 or head or golden or generic or 120 or partially or hipgraph" > gpurun_out/pytest_h.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_h.log; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_h.log | tail -8
timeout 600 python tools/tune_x3.py --tiles 210,212,214 > gpurun_out/tune_x3_h.log 2>&1; grep -v "^    \[" gpurun_out/tune_x3_h.log | cut -c1-330 | tail -32
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > gpurun_out/bench_ps.log 2>&1; grep "^{" gpurun_out/bench_ps.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('presplit', d['value'], d['sustained'], d['roofline']['families'])"
OSVOS_X3_PRESPLIT=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > gpurun_out/bench_nops.log 2>&1; grep "^{" gpurun_out/bench_nops.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('in-kernel split', d['value'], d['sustained'], d['roofline']['families'])"
