#!/bin/bash
# cycle counts (GRBM_GUI_ACTIVE / 8) of the k-loop ablations: time alone misleads, the clock follows the data
set -u
mkdir -p gpurun_out/wg9
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/wg9
export PROBE_REPS=20
cd /tmp; export TMPDIR=/tmp
for v in "0 5" "2 5" "3 5" "5 5" "0 10"; do
  set -- $v
  bin=$R/tools/native/bin/wgrad_probe; [ "$1" != "0" ] && bin=$R/tools/native/bin/wgrad_probe_abl$1
  OSVOS_WGRAD_FORM=$2 timeout 120 rocprofv3 --kernel-trace --output-format csv -d $O/pmc_a$1_f$2/p1 -o p1 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES -- $bin 12 120 214 256 256 > $O/pmc_a$1_f$2.log 2>&1
  (cd $R; python tools/pmc_summary.py gpurun_out/wg9/pmc_a$1_f$2 | grep -A2 "wgrad_bf16" | cut -c1-200 | sed "s/^/abl $1 form $2: /")
done
