#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short -p no:cacheprovider -k "bf16" > gpurun_out/pytest_bf16.log 2>&1
grep -E "passed|failed|FAILED|Error|error" gpurun_out/pytest_bf16.log | cut -c1-300 | tail -8
timeout 300 python tools/tune_conv.py --dtype bf16 --batch 12 --reps 3 > gpurun_out/tune_bf16_b12.log 2>&1; grep -v "dgrad\|wgrad" gpurun_out/tune_bf16_b12.log | cut -c1-200 | tail -25
timeout 300 python tools/tune_conv.py --dtype bf16 --batch 1 --reps 3 > gpurun_out/tune_bf16_b1.log 2>&1; grep -v "dgrad\|wgrad" gpurun_out/tune_bf16_b1.log | cut -c1-200 | tail -25
