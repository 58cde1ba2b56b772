#!/bin/bash
# One parameterised runner for gpurun calls (replaces the per-experiment gpu_r02_*.sh scripts): runs the named steps in order, every step
# under its own timeout, logs under gpurun_out/<tag>/.   usage: tools/gpu_run.sh <tag> <step> [<step> ...]
#   steps: pytest[:expr] | smoke | bench[:args] | prof[:args] | layers:<mode batch> | cmd:<shell command>
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd $R
python -c "import torch; print(torch.__version__, torch.cuda.get_device_name(0))" > $O/env.log 2>&1
nproc >> $O/env.log; lscpu | grep "Model name" >> $O/env.log
for step in "$@"; do
  name=${step%%:*}; arg=""; [[ "$step" == *:* ]] && arg=${step#*:}
  t0=$(date +%s)
  case $name in
    pytest) timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=25 ${arg:+-k "$arg"} > $O/pytest_gpu.log 2>&1; echo "exit $?" >> $O/pytest_gpu.log; grep -E "passed|failed|FAILED|Error" $O/pytest_gpu.log | tail -25 ;;
    smoke) timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log ;;
    bench) timeout 900 python bench.py $arg > $O/bench.log 2>$O/bench.err; tail -c 3000 $O/bench.log; tail -5 $O/bench.err ;;
    prof) (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py $arg > $O/rocprof.log 2>&1); DB=$(find $O/prof -name "*.db" | head -1); python tools/prof_summary.py $DB ${PROF_STEPS:-0} $O/kernel_stats.txt "python bench.py $arg" > /dev/null 2>&1; head -40 $O/kernel_stats.txt ;;
    layers) # per-layer table (tools/layer_table.py): arg = "bf16 12" or "x3 1" (+ " --all")
       set -- $arg; M=$1; B=$2; X=${3:-}
       timeout 300 python tools/layer_table.py run --mode $M --batch $B $X | tail -1 > $O/layers_$M.json
       (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/layers_pmc_$M -o p --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python $R/tools/layer_table.py run --mode $M --batch $B $X > $O/layers_pmc_$M.log 2>&1)
       python tools/layer_table.py report $O/layers_pmc_$M $O/layers_$M.json > $O/layer_table_$M.txt 2>&1; cat $O/layer_table_$M.txt ;;
    cmd) timeout 900 bash -c "$arg" > $O/cmd_$t0.log 2>&1; tail -40 $O/cmd_$t0.log ;;
    *) echo "unknown step $step" ;;
  esac
  echo "== $step: $(( $(date +%s) - t0 )) s"
done
