#!/bin/bash
# quick GPU check: net-level parity tests + bench (+ optional extra command)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_net.py tests/test_gpu_ops.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_quick.log 2>&1
tail -3 gpurun_out/pytest_quick.log
timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/bench_quick.log 2>&1; tail -1 gpurun_out/bench_quick.log
OSVOS_TWO_STREAMS=0 timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/bench_quick_1s.log 2>&1; tail -1 gpurun_out/bench_quick_1s.log | cut -c1-200
timeout 300 python tools/tune_wgrad.py > gpurun_out/tune_wgrad.log 2>&1; cut -c1-110 gpurun_out/tune_wgrad.log | tail -8
