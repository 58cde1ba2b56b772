#!/bin/bash
# HBM traffic per step of the real training step (tools/pmc_step_traffic.py): usage: gpu_pmc_step.sh <tag> <label> <bench args...>
TAG=$1; LABEL=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
COMMON="--no-extra --no-cpu-baseline --no-prof --min-seconds 0 --settle-seconds 0.15 --steps 10 --warmup 5"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/fetch -o f --pmc FETCH_SIZE -- python $R/bench.py $COMMON "$@" > $O/fetch.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/write -o w --pmc WRITE_SIZE -- python $R/bench.py $COMMON "$@" > $O/write.log 2>&1)
cd $R; python tools/pmc_step_traffic.py $O $O/traffic.json "$LABEL" > $O/traffic.txt 2>&1; head -c 2500 $O/traffic.txt
rm -rf $O/fetch $O/write
