#!/bin/bash
# f32x3 convolution (X10, pre-split weights, conv3_2 shape, batch 1): K-loop ablations, elapsed CYCLES from PMC (zero activations for ablation 0z: the clock)
set -u
mkdir -p gpurun_out/cx3
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/cx3
cd /tmp; export TMPDIR=/tmp
for a in 0 1 2 3 4 5; do
  bin=$R/tools/native/bin/conv_probe; [ "$a" != "0" ] && bin=$R/tools/native/bin/conv_probe_x3abl$a
  timeout 120 rocprofv3 --kernel-trace --output-format csv -d $O/pmc_a$a/p1 -o p1 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES -- $bin x3ps 1 120 214 256 256 110 > $O/pmc_a$a.log 2>&1
  (cd $R; python tools/pmc_summary.py gpurun_out/cx3/pmc_a$a | grep -A3 "conv3x3_f32x3" | cut -c1-260 | sed "s/^/abl $a: /")
done
