#!/bin/bash
set -u
export TMPDIR=/tmp
run() { env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32x3 b1 $*:', d['value'], d['sustained']['value'], d['roofline']['families'])"; }
run OSVOS_SIDE_STREAM=1
run OSVOS_SIDE_STREAM=0
run OSVOS_SIDE_STREAM=0 OSVOS_THREE_STREAMS=0
run OSVOS_THREE_STREAMS=0
run OSVOS_SIDE_STREAM=1
runb() { env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --mode parent --precision bf16 --batch 12 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 b12 $*:', d['value'], d['sustained']['value'], d['roofline']['families'])"; }
runb OSVOS_SIDE_STREAM=0
runb OSVOS_SIDE_STREAM=0 OSVOS_TWO_STREAMS=1 OSVOS_THREE_STREAMS=0
