#!/bin/bash
# Second GPU pass over 'fp32h2': sign-flip probe, per-tensor gradient errors, wide-tile A/B (two workgroups per CU with two-piece LDS), kernel stats
cd "$(dirname "$0")/.."
O=gpurun_out/h2; mkdir -p $O
timeout 600 python tools/h2_flip_probe.py > $O/flips.txt 2>&1; tail -25 $O/flips.txt
timeout 900 python tools/grad_error_table.py > $O/grad_table.txt 2>&1; tail -8 $O/grad_table.txt
b() { timeout 300 env $1 python bench.py --precision $2 --no-extra --no-cpu-baseline --no-parity --steps 30 --warmup 5 --full-line 2>/dev/null | tail -1 | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('$1 $2', l['value'], l['ms_per_step'], l.get('sustained',{}).get('value'))"; }
for r in 1 2; do
  b OSVOS_X3_WIDE_TILE=10 fp32h2; b OSVOS_X3_WIDE_TILE=12 fp32h2; b OSVOS_X3_WIDE_TILE=16 fp32h2
  b OSVOS_X3_WIDE_TILE=10 fp32x3b2; b OSVOS_X3_WIDE_TILE=12 fp32x3b2
done
R=$(pwd); export TMPDIR=/tmp
for T in fp32h2 fp32x3b2; do
  A="--no-extra --no-cpu-baseline --no-parity --no-prof --min-seconds 0 --settle-seconds 1.0 --steps 20 --warmup 5 --precision $T"
  (cd /tmp && timeout 180 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$T -o bench -- python $R/bench.py $A > $R/$O/rocprof_$T.log 2>&1)
  DB=$(find $O/prof_$T -name "*.db" | head -1)
  python tools/prof_summary.py $DB 0 $O/kernel_stats_$T.txt "python bench.py $A" > /dev/null 2>&1
  python tools/step_timeline.py $DB 60 > $O/timeline_$T.txt 2>&1
  rm -rf $O/prof_$T
  head -32 $O/kernel_stats_$T.txt | cut -c1-140
done
