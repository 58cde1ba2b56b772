#!/bin/bash
# bf16 weight gradient, LDS-DMA form: fragment prefetch depth and s_setprio variants (OSVOS_WGRAD_FORM 5 / 6 / 7 / 8) against the pixel-major form 3
set -u
mkdir -p gpurun_out/wg6
O=$GRAFT_REPO_ROOT/gpurun_out/wg6
export PROBE_REPS=30
for rep in 1 2; do
for f in 3 5 7 8 9; do
  echo "== OSVOS_WGRAD_FORM=$f" >> $O/probe.txt
  OSVOS_WGRAD_FORM=$f timeout 120 tools/native/bin/wgrad_probe 12 120 214 256 256 >> $O/probe.txt 2>&1
  OSVOS_WGRAD_FORM=$f timeout 120 tools/native/bin/wgrad_probe 12 240 427 128 128 >> $O/probe.txt 2>&1
done
done
grep -E "==|kernel|checksum" $O/probe.txt | cut -c1-160
