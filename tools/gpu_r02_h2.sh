#!/bin/bash
# conv1_1's weight gradient on the main stream at the end of the backward: gradient parity subset, DP self-check (gradient-ready events), headline bench
set -u
mkdir -p gpurun_out/h2
O=$GRAFT_REPO_ROOT/gpurun_out/h2
timeout 400 python -m pytest tests/test_gpu_net.py -q -x -k "gradients_match_reference_golden or sgd_trajectory or partially_frozen or conv1_1_weight" -p no:cacheprovider > $O/tests.log 2>&1; tail -2 $O/tests.log
DP_TIME=0 DP_H=240 DP_W=427 timeout 200 python tools/dp_selfcheck.py > $O/dp.log 2>&1; grep -E "identical|DIFFERENT|NaN" $O/dp.log | head -5
for rep in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_$rep.log 2>&1
  echo "rep $rep: $(tail -1 $O/bench_$rep.log | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["sustained"]["value"])')"
done
