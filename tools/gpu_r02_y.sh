#!/bin/bash
# bf16 weight gradient: ping-pong form (OSVOS_WGRAD_FORM=10) against 3 and 5: bit-identity test, timings, clock / pipe occupancy
set -u
mkdir -p gpurun_out/wg8
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/wg8
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "forms_are_bit_identical" -p no:cacheprovider > $O/forms_test.log 2>&1; tail -2 $O/forms_test.log
export PROBE_REPS=30
for rep in 1 2; do
for f in 3 5 10; do
  echo "== OSVOS_WGRAD_FORM=$f" >> $O/probe.txt
  OSVOS_WGRAD_FORM=$f timeout 120 tools/native/bin/wgrad_probe 12 120 214 256 256 >> $O/probe.txt 2>&1
  OSVOS_WGRAD_FORM=$f timeout 120 tools/native/bin/wgrad_probe 12 240 427 128 128 >> $O/probe.txt 2>&1
  OSVOS_WGRAD_FORM=$f timeout 120 tools/native/bin/wgrad_probe 12 60 107 512 512 >> $O/probe.txt 2>&1
done
done
grep -E "==|reduce|checksum" $O/probe.txt | cut -c1-150
cd /tmp
for f in 10; do
  OSVOS_WGRAD_FORM=$f timeout 120 rocprofv3 --kernel-trace --output-format csv -d $O/pmc_f$f/p1 -o p1 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES -- $R/tools/native/bin/wgrad_probe 12 120 214 256 256 > $O/pmc_f$f.log 2>&1
  (cd $R; python tools/pmc_summary.py gpurun_out/wg8/pmc_f$f | grep -A4 "wgrad_bf16" | cut -c1-220 | sed "s/^/form $f: /")
done
