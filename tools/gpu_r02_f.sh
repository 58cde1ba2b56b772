#!/bin/bash
# Round-2 call F: full GPU tier (gradient-ready events, overlapped all-reduce self-check, device-augment scripts), default bench line.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_gpu.log | tail -20
DP_H=240 DP_W=427 timeout 300 python tools/dp_selfcheck.py > gpurun_out/dp_selfcheck.log 2>&1; tail -8 gpurun_out/dp_selfcheck.log
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | cut -c1-400
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --force-dist > gpurun_out/bench_force_dist.log 2>&1; tail -1 gpurun_out/bench_force_dist.log | cut -c1-300
