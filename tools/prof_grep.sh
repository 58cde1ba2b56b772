#!/bin/bash
# rocprofv3 kernel stats of one bench.py command, filtered: tools/prof_grep.sh <out tag> "<bench args>" <grep pattern> [steps]
TAG=$1; ARGS=$2; PAT=$3; STEPS=${4:-8}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
(cd /tmp && TMPDIR=/tmp timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra --min-seconds 0 $ARGS > $O/rocprof.log 2>&1)
DB=$(find $O/prof -name "*.db" | head -1)
python $R/tools/prof_summary.py $DB $STEPS $O/kernel_stats.txt "python bench.py $ARGS" > /dev/null 2>&1
grep -E "$PAT|total kernel" $O/kernel_stats.txt | cut -c1-170
