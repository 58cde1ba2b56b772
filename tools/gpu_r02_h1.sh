#!/bin/bash
# merged head backward + last side branch on the main stream: parity tests, then the headline A/B (OSVOS_HEAD_BWD_MERGED=0 / 1), interleaved
set -u
mkdir -p gpurun_out/h1
O=$GRAFT_REPO_ROOT/gpurun_out/h1
timeout 900 python -m pytest tests/test_gpu_net.py tests/test_gpu_baseline_configs.py tests/test_gpu_ops.py -q -x -k "not forms" -p no:cacheprovider > $O/tests.log 2>&1; tail -2 $O/tests.log
for rep in 1 2; do
for m in 0 1; do
  OSVOS_HEAD_BWD_MERGED=$m timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_m${m}_$rep.log 2>&1
  echo "merged=$m rep $rep: $(tail -1 $O/bench_m${m}_$rep.log | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["sustained"]["value"])')"
done
done
OSVOS_HEAD_BWD_MERGED=1 timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extra --mode parent --precision bf16 --batch 12 > $O/bench_bf16.log 2>&1
echo "bf16 b12: $(tail -1 $O/bench_bf16.log | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["sustained"]["value"])')"
