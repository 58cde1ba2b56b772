#!/bin/bash
# bf16 weight gradient, LDS-DMA form: issue patterns (OSVOS_WGRAD_FORM 4 / 5 / 6) and the data-less DMA ablation
set -u
mkdir -p gpurun_out/wg2
O=$GRAFT_REPO_ROOT/gpurun_out/wg2
for rep in 1 2; do
for f in 3 4 5 6; do
  echo "== OSVOS_WGRAD_FORM=$f" >> $O/probe.txt
  OSVOS_WGRAD_FORM=$f timeout 120 tools/native/bin/wgrad_probe 12 120 214 256 256 >> $O/probe.txt 2>&1
  OSVOS_WGRAD_FORM=$f timeout 120 tools/native/bin/wgrad_probe 12 240 427 128 128 >> $O/probe.txt 2>&1
done
done
echo "== abl5 OSVOS_WGRAD_FORM=4" >> $O/probe.txt
OSVOS_WGRAD_FORM=4 timeout 60 tools/native/bin/wgrad_probe_abl5 12 120 214 256 256 >> $O/probe.txt 2>&1
grep -E "==|kernel" $O/probe.txt | cut -c1-160
