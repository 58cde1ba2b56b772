#!/bin/bash
# clock and matrix-pipe occupancy of the bf16 weight-gradient forms (same work, different schedules): is the product constant?
set -u
mkdir -p gpurun_out/wg7
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/wg7
export PROBE_REPS=30
cd /tmp; export TMPDIR=/tmp
for f in 3 5 9 6; do
  OSVOS_WGRAD_FORM=$f timeout 120 rocprofv3 --kernel-trace --output-format csv -d $O/pmc_f$f/p1 -o p1 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES -- $R/tools/native/bin/wgrad_probe 12 120 214 256 256 > $O/pmc_f$f.log 2>&1
  (cd $R; python tools/pmc_summary.py gpurun_out/wg7/pmc_f$f | grep -A4 "wgrad_bf16" | cut -c1-220 | sed "s/^/form $f: /")
done
