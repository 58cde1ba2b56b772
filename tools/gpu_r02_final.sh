#!/bin/bash
# Round-2 final validation: full GPU tier, smoke, default bench line (headline + nested configs), kernel trace of the headline, DP self-check.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch; print(torch.__version__, torch.cuda.get_device_name(0))" > gpurun_out/env.log 2>&1
nproc >> gpurun_out/env.log; lscpu | grep "Model name" >> gpurun_out/env.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_gpu.log | tail -12
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | cut -c1-300
DP_TIME=0 DP_H=240 DP_W=427 timeout 300 python tools/dp_selfcheck.py > gpurun_out/dp_selfcheck.log 2>&1; tail -1 gpurun_out/dp_selfcheck.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_final -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --min-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/rocprof_final.log 2>&1
cd $GRAFT_REPO_ROOT; ls gpurun_out/prof_final | head -3
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_bf16 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra --min-seconds 0 --mode parent --precision bf16 --batch 12 > $GRAFT_REPO_ROOT/gpurun_out/rocprof_bf16.log 2>&1
cd $GRAFT_REPO_ROOT; ls gpurun_out/prof_bf16 | head -3
