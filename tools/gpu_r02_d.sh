#!/bin/bash
# Round-2 call D: fp32x3 is the module default -- full GPU tier, bench default line (+ extras), hipGraph micro-batch replay, 1080p sweep.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_gpu.log | tail -20
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | cut -c1-3000
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --graph-train 1 > gpurun_out/bench_graph.log 2>&1; tail -2 gpurun_out/bench_graph.log | cut -c1-900
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --graph-train 1 --precision fp32 > gpurun_out/bench_graph_exact.log 2>&1; tail -1 gpurun_out/bench_graph_exact.log | cut -c1-300
timeout 300 python tools/tune_x3.py --height 1080 --width 1920 --batch 4 --tiles 210,212,214 --reps 2 > gpurun_out/tune_x3_1080p.log 2>&1; grep -v "^    \[" gpurun_out/tune_x3_1080p.log | tail -30
