#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for ts in 1 0; do
OSVOS_TWO_STREAMS=$ts timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --mode parent --precision bf16 --batch 12 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 parent b12 two_streams=$ts:', d['value'], d['sustained']['value'], d['roofline']['families'])"
done
OSVOS_THREE_STREAMS=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --mode parent --precision bf16 --batch 12 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 parent b12 three_streams=0:', d['value'], d['sustained']['value'], d['roofline']['families'])"
for ts in 0; do
OSVOS_TWO_STREAMS=$ts timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32x3 b1 two_streams=$ts:', d['value'], d['sustained']['value'], d['roofline']['families'])"
done
OSVOS_THREE_STREAMS=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32x3 b1 three_streams=0:', d['value'], d['sustained']['value'], d['roofline']['families'])"
