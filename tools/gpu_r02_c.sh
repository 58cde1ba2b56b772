#!/bin/bash
# Round-2 call C: f32x3 weight gradient + measured tile rule + side_prep split-K: op tests, sweep, whole-net parity, bench, trace.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short -p no:cacheprovider -s -k "f32x3 or splitk or conv3x3_forward_all_tiles" > gpurun_out/pytest_x3_ops.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_x3_ops.log; grep -E "passed|failed|FAILED|vs float64" gpurun_out/pytest_x3_ops.log | tail -12
timeout 600 python tools/tune_x3.py --tiles 210,212,214 > gpurun_out/tune_x3_c.log 2>&1; grep -v "^    \[" gpurun_out/tune_x3_c.log | tail -32
timeout 900 python -m pytest tests/test_gpu_net.py -m gpu -q --tb=short -p no:cacheprovider -s -k "golden or hipgraph or intermediate or 120" > gpurun_out/pytest_x3_net.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_x3_net.log; grep -E "passed|failed|FAILED|Error|gradients \(ours" gpurun_out/pytest_x3_net.log | cut -c1-300 | tail -12
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --precision fp32x3 > gpurun_out/bench_x3.log 2>&1; tail -1 gpurun_out/bench_x3.log | cut -c1-1200
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > gpurun_out/bench_exact.log 2>&1; tail -1 gpurun_out/bench_exact.log | cut -c1-700
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_x3 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --min-seconds 0 --precision fp32x3 > $GRAFT_REPO_ROOT/gpurun_out/rocprof_x3.log 2>&1
cd $GRAFT_REPO_ROOT; ls gpurun_out/prof_x3 | head
