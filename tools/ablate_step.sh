#!/bin/bash
# What each piece of the backward costs at step level: bench.py with OSVOS_DBG_SKIP masks (csrc/net.cpp; results are WRONG, timing only).
# Needs a probe build of the library: (cd osvos-pytorch_amd/csrc && make clean && make EXTRA=-DOSVOS_DBG_ABLATIONS)
# usage: tools/ablate_step.sh "<bench args>" mask [mask ...]
ARGS=$1; shift
for m in "$@"; do
  v=$(OSVOS_DBG_SKIP=$m python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --min-seconds 1 $ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%.1f fps  %.3f ms/step  (sustained %.1f)' % (d['value'], d['ms_per_step'], d.get('sustained', {}).get('value', 0)))")
  echo "skip=$m  $v"
done
