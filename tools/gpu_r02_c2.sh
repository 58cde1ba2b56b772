#!/bin/bash
# f32x3 convolution: the same launch on noise / ReLU-like / zero activations (is it the clock?), with the clock from PMC
set -u
mkdir -p gpurun_out/cx2
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/cx2/probe.txt
P=tools/native/bin/conv_probe
for d in noise relu zero; do echo "== PROBE_DATA=$d 1 120 214 256 256" >> $O; PROBE_DATA=$d timeout 120 $P x3ps 1 120 214 256 256 110,116 >> $O 2>&1; done
cat $O
cd /tmp; export TMPDIR=/tmp
for d in noise relu zero; do
  PROBE_DATA=$d timeout 120 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/cx2/pmc_$d/p1 -o p1 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES -- $R/$P x3ps 1 120 214 256 256 110 > $R/gpurun_out/cx2/pmc_$d.log 2>&1
  (cd $R; python tools/pmc_summary.py gpurun_out/cx2/pmc_$d | grep -A3 "conv3x3_f32x3" | cut -c1-200 | sed "s/^/$d: /")
done
