#!/bin/bash
# quick step-level reading of the tree as built: fp32x3 and fp32x3b2, three runs each
cd "$(dirname "$0")/.."
b() { timeout 300 python bench.py --precision $1 --no-extra --no-cpu-baseline --no-parity --steps 40 --warmup 5 --full-line 2>/dev/null | tail -1 | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('$1', l['value'], l['ms_per_step'], l.get('sustained',{}).get('value'), l['roofline'].get('pipe_sustained',{}).get('noise'))"; }
for r in 1 2 3; do b fp32x3; b fp32x3b2; done
