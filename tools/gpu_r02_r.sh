#!/bin/bash
# bf16 weight gradient: LDS-DMA form (OSVOS_WGRAD_FORM=4/5) against the pixel-major default (3): bit-identity, native probe timings,
# k-loop ablations, PMC counters.
set -u
mkdir -p gpurun_out/wg
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/wg
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "forms_are_bit_identical" -p no:cacheprovider > $O/forms_test.log 2>&1; tail -3 $O/forms_test.log
for f in 3 4 5; do
  echo "== OSVOS_WGRAD_FORM=$f" >> $O/probe.txt
  OSVOS_WGRAD_FORM=$f timeout 120 tools/native/bin/wgrad_probe >> $O/probe.txt 2>&1
done
for abl in 1 2 3; do
  for f in 3 4; do
    echo "== abl$abl OSVOS_WGRAD_FORM=$f" >> $O/abl.txt
    OSVOS_WGRAD_FORM=$f timeout 60 tools/native/bin/wgrad_probe_abl$abl 12 120 214 256 256 >> $O/abl.txt 2>&1
  done
done
echo "== abl4 OSVOS_WGRAD_FORM=3" >> $O/abl.txt
OSVOS_WGRAD_FORM=3 timeout 60 tools/native/bin/wgrad_probe_abl4 12 120 214 256 256 >> $O/abl.txt 2>&1
grep -E "==|kernel" $O/probe.txt $O/abl.txt | cut -c1-160
cd /tmp
for f in 3 4; do
  OSVOS_WGRAD_FORM=$f timeout 120 rocprofv3 --kernel-trace --output-format csv -d $O/pmc_f$f/p1 -o p1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- $R/tools/native/bin/wgrad_probe 12 120 214 256 256 > $O/pmc_f${f}_p1.log 2>&1
  OSVOS_WGRAD_FORM=$f timeout 120 rocprofv3 --kernel-trace --output-format csv -d $O/pmc_f$f/p2 -o p2 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -- $R/tools/native/bin/wgrad_probe 12 120 214 256 256 > $O/pmc_f${f}_p2.log 2>&1
  OSVOS_WGRAD_FORM=$f timeout 120 rocprofv3 --kernel-trace --output-format csv -d $O/pmc_f$f/p4 -o p4 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM -- $R/tools/native/bin/wgrad_probe 12 120 214 256 256 > $O/pmc_f${f}_p4.log 2>&1
  cd $R; python tools/pmc_summary.py gpurun_out/wg/pmc_f$f > $O/pmc_f${f}_summary.txt 2>&1; cd /tmp
done
cat $O/pmc_f3_summary.txt $O/pmc_f4_summary.txt | cut -c1-250 | head -40
