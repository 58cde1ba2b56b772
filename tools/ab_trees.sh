#!/bin/bash
# same-box A/B of two whole TREES (library + host code + bench.py of each): tools/ab_trees.sh <treeA> <treeB> <rounds> <bench args...>
# e.g. the round-3 tree against HEAD:   git archive e7e6690 ... | tar -x -C _ab/r03 ; make -C _ab/r03/osvos-pytorch_amd/csrc ;
#   tools/ab_trees.sh _ab/r03 . 2 --steps 50 --warmup 10        (alternates A, B, A, B in one process sequence on one box)
A=$1; B=$2; R=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for rnd in $(seq 1 $R); do
  for T in $A $B; do
    X=""; grep -q -- "--no-parity" $ROOT/$T/bench.py && X="--no-parity"      # (trees older than round 5 have no parity gate)
    v=$(cd $ROOT/$T && OSVOS_AUTOBUILD=0 python bench.py --no-extra --no-cpu-baseline $X --min-seconds 2 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%.1f fps  %.3f ms/step  (sustained %.1f)' % (d['value'], d['ms_per_step'], (d.get('sustained') or {}).get('value', 0)))")
    echo "[$T]  $v"
  done
done
