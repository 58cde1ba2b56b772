#!/bin/bash
# one-rank cost of the data-parallel exchange paths on the headline workload (bench.py --force-dist: the collective runs with one rank)
run() {
  v=$(env $1 python bench.py --steps 40 --warmup 20 --no-extra --no-cpu-baseline --min-seconds 1 $2 2>/tmp/ab_dp.err | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%.1f fps  (sustained %.1f)  ranks_seen %s  %s' % (d['value'], d.get('sustained', {}).get('value', 0), d['config']['rccl_ranks_seen'], d['config']['grad_allreduce']))")
  echo "[$1 $2]  $v"; [ -z "$v" ] && tail -5 /tmp/ab_dp.err
}
for rnd in 1 2; do
  run "A=1" ""
  run "A=1" "--force-dist"
  run "OSVOS_DP_BACKEND=abi" "--force-dist"
  run "OSVOS_DP_BACKEND=abi OSVOS_DP_OVERLAP=1" "--force-dist"
done
