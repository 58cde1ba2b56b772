#!/bin/bash
set -u
export TMPDIR=/tmp
for i in 1 2 3; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no dist run $i:', d['value'], d['sustained']['value'], d['timed_region_detail'])"
done
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --force-dist 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('force-dist run $i:', d['value'], d['sustained']['value'], d['timed_region_detail'])"
done
