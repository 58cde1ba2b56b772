#!/bin/bash
# Round-end evidence in one gpurun call: the default bench line, then settled rocprofv3 kernel-trace summaries + per-step timelines of configs[1]
# and configs[2] (what profiles/r04_bench_default.json, r04_bench_*_kernel_stats.txt and r04_step_timeline*.txt are).  usage: gpu_evidence.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp; O=$R/gpurun_out/${1:-evidence}; mkdir -p $O
t0=$(date +%s)
[ -z "$NOBENCH" ] && { timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $? $(( $(date +%s) - t0 )) s" > $O/times.txt; }
prof() { # tag, bench args...
  T=$1; shift
  A="--no-extra --no-cpu-baseline --no-parity --no-prof --min-seconds 0 --settle-seconds 1.0 --steps 20 --warmup 5 $*"
  (cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats -d $O/prof_$T -o bench -- python $R/bench.py $A > $O/rocprof_$T.log 2>&1)
  DB=$(find $O/prof_$T -name "*.db" | head -1)
  python tools/prof_summary.py $DB 0 $O/kernel_stats_$T.txt "python bench.py $A" > /dev/null 2>&1      # 0: count the profiled steps
  python tools/step_timeline.py $DB 60 > $O/timeline_$T.txt 2>&1
  rm -rf $O/prof_$T
  echo "prof $T $(( $(date +%s) - t0 )) s" >> $O/times.txt
}
prof configs1
prof configs2 --mode parent --precision bf16 --batch 12
prof configs1_fp32x3b2 --precision fp32x3b2
prof configs1_fp32h2 --precision fp32h2
cat $O/times.txt
