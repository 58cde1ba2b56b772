#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/h2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "fp16_pairs or two_pieces" > $O/ops4.txt 2>&1; tail -3 $O/ops4.txt
b() { timeout 300 env $1 python bench.py --precision $2 --no-extra --no-cpu-baseline --no-parity --steps 30 --warmup 5 --full-line 2>/dev/null | tail -1 | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('$1 $2', l['value'], l['ms_per_step'], l.get('sustained',{}).get('value'))"; }
for r in 1 2 3; do
  for p in fp32x3b2 fp32h2; do
    b A=0 $p; b OSVOS_X2_TILE_FOR_10=11 $p; b OSVOS_X2_TILE_FOR_12=13 $p
  done
done
