#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py -m gpu -q --tb=short -p no:cacheprovider -s -k "bf16" > gpurun_out/pytest_bf16.log 2>&1
grep -E "passed|failed|FAILED|Error|error" gpurun_out/pytest_bf16.log | cut -c1-300 | tail -8
timeout 300 python bench.py --precision bf16 --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/bench_bf16.log 2>&1; tail -1 gpurun_out/bench_bf16.log | cut -c1-200
timeout 300 python tools/tune_conv.py --dtype bf16 --batch 1 > gpurun_out/tune_bf16_b1.log 2>&1; grep -v "dgrad" gpurun_out/tune_bf16_b1.log | cut -c1-200 | tail -45
timeout 300 python tools/tune_conv.py --dtype bf16 --batch 12 --reps 3 > gpurun_out/tune_bf16_b12.log 2>&1; grep -v "dgrad" gpurun_out/tune_bf16_b12.log | cut -c1-200 | tail -45
