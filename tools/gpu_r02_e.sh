#!/bin/bash
# Round-2 call E: skinny f32x3 tiles, PMC passes on the f32x3 kernels, default bench line with extras in their own processes, kernel trace.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py -m gpu -q --tb=short -p no:cacheprovider -s -k "f32x3 or golden or partially" > gpurun_out/pytest_e.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_e.log; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_e.log | tail -8
bash tools/gpu_pmc_x3.sh > gpurun_out/pmc_x3.log 2>&1; tail -40 gpurun_out/pmc_x3.log | cut -c1-250
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | cut -c1-600
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_e -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --min-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/rocprof_e.log 2>&1
cd $GRAFT_REPO_ROOT; ls gpurun_out/prof_e | head -3
