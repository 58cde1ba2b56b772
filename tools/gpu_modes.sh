#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_net.py -m gpu -q --tb=short -p no:cacheprovider -x -k "hipgraph or golden" > gpurun_out/pytest_modes.log 2>&1; tail -3 gpurun_out/pytest_modes.log
timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/bench_online.log 2>&1; tail -1 gpurun_out/bench_online.log | cut -c1-420
timeout 300 python bench.py --mode infer --height 1080 --width 1920 --batch 4 --graph 1 --steps 20 --warmup 5 > gpurun_out/bench_infer1080.log 2>&1; tail -1 gpurun_out/bench_infer1080.log
timeout 300 python bench.py --mode infer --height 1080 --width 1920 --batch 4 --graph 0 --steps 20 --warmup 5 > gpurun_out/bench_infer1080_eager.log 2>&1; tail -1 gpurun_out/bench_infer1080_eager.log | cut -c1-300
timeout 600 python bench.py --mode parent --batch 12 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_parent_b12.log 2>&1; tail -1 gpurun_out/bench_parent_b12.log
