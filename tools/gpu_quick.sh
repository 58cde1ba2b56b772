#!/bin/bash
# quick GPU check: parity tests + bench (+ split-K sweep)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_net.py tests/test_gpu_ops.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_quick.log 2>&1
tail -3 gpurun_out/pytest_quick.log
timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/bench_quick.log 2>&1; tail -1 gpurun_out/bench_quick.log | cut -c1-1000
for ks in 1 2 4; do OSVOS_CONV_KSPLIT=$ks timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-160; done
