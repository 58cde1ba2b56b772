#!/bin/bash
# side_prep weight gradients on the third stream: gradient parity subset, DP self-check, headline A/B (OSVOS_SIDE_WGRAD_AUX2=0 / 1)
set -u
mkdir -p gpurun_out/h3
O=$GRAFT_REPO_ROOT/gpurun_out/h3
timeout 300 python -m pytest tests/test_gpu_net.py -q -x -k "gradients_match_reference_golden or sgd_trajectory or partially_frozen" -p no:cacheprovider > $O/tests.log 2>&1; tail -1 $O/tests.log
DP_TIME=0 DP_H=240 DP_W=427 timeout 200 python tools/dp_selfcheck.py > $O/dp.log 2>&1; grep -E "identical|DIFFERENT" $O/dp.log | head -3
for m in 0 1 0 1; do
  OSVOS_SIDE_WGRAD_AUX2=$m timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --min-seconds 1 > $O/bench_$m.log 2>&1
  echo "aux2=$m: $(tail -1 $O/bench_$m.log | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["sustained"]["value"])')"
done
