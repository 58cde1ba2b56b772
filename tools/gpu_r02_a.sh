#!/bin/bash
# Round-2 call A: full GPU parity tier, smoke, default bench line (with extra_configs), rocprofv3 kernel trace of the headline.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch; print(torch.__version__, torch.cuda.get_device_name(0))" > gpurun_out/env.log 2>&1
nproc >> gpurun_out/env.log; lscpu | grep "Model name" >> gpurun_out/env.log
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_gpu.log | tail -30
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --min-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT; ls -R gpurun_out/prof | head -20
