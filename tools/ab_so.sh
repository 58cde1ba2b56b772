#!/bin/bash
# same-box A/B of BUILDS of libosvos_hip.so (probe builds with -D switches kept under ab_so/<name>.so, git-ignored, shipped by gpurun): alternating fresh
# processes, step level.  usage: tools/ab_so.sh "<precision> [<precision> ..]" <rounds> <name> [<name> ..]
cd "$(dirname "$0")/.."
PRECS=$1; ROUNDS=$2; shift; shift
cp osvos-pytorch_amd/libosvos_hip.so /tmp/keep.so
b() { timeout 300 python bench.py --precision $2 --no-extra --no-cpu-baseline --no-parity --steps 40 --warmup 5 --full-line 2>/dev/null | tail -1 | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('%-10s %-9s' % ('$1', '$2'), l['value'], l['ms_per_step'], l.get('sustained',{}).get('value'))"; }
for r in $(seq $ROUNDS); do
  for p in $PRECS; do
    for n in "$@"; do cp ab_so/$n.so osvos-pytorch_amd/libosvos_hip.so; OSVOS_AUTOBUILD=0 b $n $p; done
  done
done
cp /tmp/keep.so osvos-pytorch_amd/libosvos_hip.so
