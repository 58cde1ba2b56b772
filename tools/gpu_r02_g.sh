#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for ov in 1 0; do
OSVOS_DP_OVERLAP=$ov timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --force-dist 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('force-dist overlap=$ov', d['value'], d['sustained'])"
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no dist', d['value'], d['sustained'])"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_fd -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --min-seconds 0 --force-dist > $GRAFT_REPO_ROOT/gpurun_out/rocprof_fd.log 2>&1
cd $GRAFT_REPO_ROOT; ls gpurun_out/prof_fd | head -3
