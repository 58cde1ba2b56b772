#!/bin/bash
set -u
export TMPDIR=/tmp
run() { env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --force-dist 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('force-dist $*:', d['value'], d['sustained']['value'])"; }
run OSVOS_DP_BACKEND=abi GPU_MAX_HW_QUEUES=16
run OSVOS_DP_BACKEND=abi GPU_MAX_HW_QUEUES=4
run OSVOS_DP_BACKEND=torch GPU_MAX_HW_QUEUES=16
run OSVOS_DP_BACKEND=torch GPU_MAX_HW_QUEUES=4
run OSVOS_DP_BACKEND=torch GPU_MAX_HW_QUEUES=2
env GPU_MAX_HW_QUEUES=16 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no dist, 16 queues:', d['value'], d['sustained']['value'])"
env GPU_MAX_HW_QUEUES=4 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no dist, 4 queues:', d['value'], d['sustained']['value'])"
