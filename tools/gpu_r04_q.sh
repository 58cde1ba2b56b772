cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r04w; mkdir -p $O
t0=$(date +%s)
timeout 230 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $? $(( $(date +%s) - t0 )) s" > $O/times.txt
prof() { # tag, steps for summary, bench args...
  T=$1; shift
  (cd /tmp && timeout 90 rocprofv3 --kernel-trace --stats -d $O/prof_$T -o bench -- python $R/bench.py --no-extra --no-cpu-baseline --no-prof --min-seconds 0 --settle-seconds 1.0 --steps 20 --warmup 5 "$@" > $O/rocprof_$T.log 2>&1)
  DB=$(find $O/prof_$T -name "*.db" | head -1)
  python tools/prof_summary.py $DB 25 $O/kernel_stats_$T.txt "python bench.py --no-extra --no-cpu-baseline --no-prof --min-seconds 0 --settle-seconds 1.0 --steps 20 --warmup 5 $*" > /dev/null 2>&1
  python tools/step_timeline.py $DB 60 0 1 > $O/timeline_$T.txt 2>&1
  grep '^{' $O/rocprof_$T.log | tail -1 > $O/bench_under_rocprof_$T.json
  rm -rf $O/prof_$T
  echo "prof $T $(( $(date +%s) - t0 )) s" >> $O/times.txt
}
prof configs1
prof configs2 --mode parent --precision bf16 --batch 12
