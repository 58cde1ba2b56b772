#!/bin/bash
# native probes (tools/native/bin/, git-ignored, travels with the gpurun snapshot)
set -e
cd "$(dirname "$0")"
mkdir -p bin
C=../../osvos-pytorch_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function -DOSVOS_WGRAD_PROF \
  $C/wgrad_bf16.hip $C/wgrad_f32.hip $C/wgrad_small_f32.hip -x hip $C/errors.cpp wgrad_probe.cpp -o bin/wgrad_probe
