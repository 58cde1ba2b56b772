#!/bin/bash
# Round-2 call B: f32x3 convolution -- op tests, tile sweep, whole-net parity under fp32x3, bench in both fp32 modes.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short -p no:cacheprovider -s -k "f32x3" > gpurun_out/pytest_x3_ops.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_x3_ops.log; grep -E "passed|failed|FAILED|K=4608" gpurun_out/pytest_x3_ops.log | tail -12
timeout 600 python tools/tune_x3.py > gpurun_out/tune_x3.log 2>&1; grep -v "^    \[" gpurun_out/tune_x3.log | tail -30
timeout 900 python -m pytest tests/test_gpu_net.py tests/test_gpu_baseline_configs.py -m gpu -q --tb=short -p no:cacheprovider -s -k "fp32x3 or bf16_parent" > gpurun_out/pytest_x3_net.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_x3_net.log; grep -E "passed|failed|FAILED|Error|fp32x3 gradients|bf16 854x480" gpurun_out/pytest_x3_net.log | cut -c1-400 | tail -30
OSVOS_PRECISION=fp32x3 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > gpurun_out/bench_x3.log 2>&1; tail -1 gpurun_out/bench_x3.log | cut -c1-1500
