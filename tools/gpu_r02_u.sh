#!/bin/bash
# does the bf16 weight gradient's speed depend on the DATA (matrix-pipe power -> clock)?  same kernel, same traffic, different operand bits
set -u
mkdir -p gpurun_out/wg4
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/wg4
for d in noise relu const zero noise; do
  echo "== PROBE_DATA=$d OSVOS_WGRAD_FORM=4" >> $O/probe.txt
  PROBE_DATA=$d OSVOS_WGRAD_FORM=4 timeout 120 tools/native/bin/wgrad_probe 12 120 214 256 256 >> $O/probe.txt 2>&1
done
grep -E "==|kernel" $O/probe.txt | cut -c1-160
cd /tmp; export TMPDIR=/tmp
for d in noise relu zero; do
  PROBE_DATA=$d OSVOS_WGRAD_FORM=4 timeout 120 rocprofv3 --kernel-trace --output-format csv -d $O/pmc_$d/p1 -o p1 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -- $R/tools/native/bin/wgrad_probe 12 120 214 256 256 > $O/pmc_$d.log 2>&1
  (cd $R; python tools/pmc_summary.py gpurun_out/wg4/pmc_$d | grep -A3 "dma_kernel" | cut -c1-200 | sed "s/^/$d: /")
done
