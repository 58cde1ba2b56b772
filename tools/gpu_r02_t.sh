#!/bin/bash
# configs[2] (bf16, batch 12, parent loop) under the weight-gradient forms 3 / 4 / 5, twice each (interleaved)
set -u
mkdir -p gpurun_out/wg3
O=$GRAFT_REPO_ROOT/gpurun_out/wg3
for rep in 1 2; do
for f in 3 5 4; do
  OSVOS_WGRAD_FORM=$f timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extra --mode parent --precision bf16 --batch 12 > $O/bench_f${f}_$rep.log 2>&1
  echo "form $f rep $rep: $(tail -1 $O/bench_f${f}_$rep.log | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d.get("sustained",{}).get("value"))')"
done
done
