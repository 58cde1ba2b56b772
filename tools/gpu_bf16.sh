#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py -m gpu -q --tb=short -p no:cacheprovider -s -k "bf16" > gpurun_out/pytest_bf16.log 2>&1
grep -E "passed|failed|FAILED|bf16 gradients|Error|error" gpurun_out/pytest_bf16.log | cut -c1-700 | tail -20
timeout 300 python bench.py --precision bf16 --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/bench_bf16.log 2>&1; tail -1 gpurun_out/bench_bf16.log | cut -c1-900
timeout 300 python bench.py --precision bf16 --mode parent --batch 12 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_bf16_b12.log 2>&1; tail -1 gpurun_out/bench_bf16_b12.log | cut -c1-900
