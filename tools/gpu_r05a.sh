#!/bin/bash
# round 5, call A: full GPU tier at the merged tree, then same-box A/Bs (r03 tree vs HEAD; pool code; wave priority), then the default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp; O=$R/gpurun_out/r05a; mkdir -p $O
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=15 > $O/pytest_gpu.log 2>&1; echo "exit $?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log; echo "pytest $(( $(date +%s) - t0 )) s"
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
{ echo "== configs[1]: r03 tree (e7e6690) vs HEAD"; timeout 400 tools/ab_trees.sh _ab/r03 . 2 --steps 50 --warmup 10
  echo "== configs[2]: r03 tree vs HEAD"; timeout 500 tools/ab_trees.sh _ab/r03 . 2 --steps 30 --warmup 5 --mode parent --precision bf16 --batch 12; } > $O/ab_r03_vs_head.txt 2>&1
cat $O/ab_r03_vs_head.txt; echo "ab trees $(( $(date +%s) - t0 )) s"
{ echo "== configs[2]: OSVOS_POOL_CODE"; timeout 400 tools/ab_env.sh "--mode parent --precision bf16 --batch 12" "OSVOS_POOL_CODE=0" "OSVOS_POOL_CODE=1"; } > $O/ab_pool_code.txt 2>&1
cat $O/ab_pool_code.txt
{ echo "== configs[1]: OSVOS_X3_PRIO"; timeout 400 tools/ab_env.sh "" "OSVOS_X3_PRIO=0" "OSVOS_X3_PRIO=1" "OSVOS_X3_PRIO=2"; } > $O/ab_setprio.txt 2>&1
cat $O/ab_setprio.txt; echo "ab env $(( $(date +%s) - t0 )) s"
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; tail -c 1500 $O/bench_default.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05a/bench_default.json") if l.startswith("{")][-1])
print("headline", d["value"], d["ms_per_step"], "parity", json.dumps(d.get("parity"))[:1500])
for e in d.get("extra_configs") or []:
    print(e.get("config", "")[:40], e.get("value"), json.dumps(e.get("parity"))[:600])
PY
echo "total $(( $(date +%s) - t0 )) s"
