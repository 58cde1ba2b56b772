#!/bin/bash
# same-box A/B of environment switches at step level: tools/ab_env.sh "<bench args>" "<ENV=.. ENV=..>" ["<ENV..>" ...]   ("-" = no switch); two rounds
ARGS=$1; shift
for rnd in 1 2; do
  for e in "$@"; do
    [ "$e" == "-" ] && e=""
    v=$(env $e python bench.py --steps 40 --warmup 20 --no-extra --no-cpu-baseline --no-parity --min-seconds 1 $ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%.1f fps  %.3f ms/step  (sustained %.1f)' % (d['value'], d['ms_per_step'], d.get('sustained_value') or (d.get('sustained') or {}).get('value', 0)))")
    echo "[$e]  $v"
  done
done
