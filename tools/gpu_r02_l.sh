#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
DP_H=240 DP_W=427 timeout 300 python tools/dp_selfcheck.py > gpurun_out/dp_selfcheck.log 2>&1; grep -v "^$" gpurun_out/dp_selfcheck.log | grep -v "rccl\|RCCL\|HIP ver\|ROCm ver\|Hostname\|socket" | cut -c1-220 | tail -16
run() { env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --force-dist 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('force-dist $*:', d['value'], d['sustained']['value'])"; }
run OSVOS_DP_BACKEND=abi OSVOS_DP_OVERLAP=0
run OSVOS_DP_BACKEND=abi OSVOS_DP_OVERLAP=1
run OSVOS_DP_BACKEND=torch OSVOS_DP_OVERLAP=0
run OSVOS_DP_BACKEND=torch OSVOS_DP_OVERLAP=1
