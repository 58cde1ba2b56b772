#!/bin/bash
# f32x3 convolution, batch 1: the 512-pixel x 64-cout tiles (X16 / X17) against the production tiles, pre-split weights, native probe
set -u
mkdir -p gpurun_out/cx
O=$GRAFT_REPO_ROOT/gpurun_out/cx/probe.txt
P=tools/native/bin/conv_probe
run() { echo "== $*" >> $O; timeout 120 $P x3ps $* >> $O 2>&1; }
run 1 120 214 256 256 110,10,112,116,16,117
run 1 240 427 128 128 110,10,112,116,16
run 1 480 854 64 64 112,12,116,16
run 1 240 427 64 128 110,112,116
run 1 120 214 128 256 110,112,116
run 1 60 107 512 512 114,14,117,17,116
PROBE_KSPLIT=2 run 1 60 107 512 512 114,117,116
run 1 60 107 256 512 114,117,116
cat $O
