#!/bin/bash
# native probes (tools/native/bin/, git-ignored, travels with the gpurun snapshot)
# usage: build.sh [ABL]   ABL = 1 no MFMA, 2 no LDS fragment reads, 3 no vmem / DMA inside the k-loop, 4 no ds_write per patch (timing ablations, wrong results)
set -e
cd "$(dirname "$0")"
mkdir -p bin
C=../../osvos-pytorch_amd/csrc
ABL=${1:-0}
OUT=bin/wgrad_probe; [ "$ABL" != "0" ] && OUT=bin/wgrad_probe_abl$ABL
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function -DOSVOS_WGRAD_PROF -DOSVOS_WGRAD_ALL_FORMS -DOSVOS_WGRAD_ABL=$ABL \
  $C/wgrad_bf16.hip $C/wgrad_f32.hip $C/wgrad_small_f32.hip -x hip $C/errors.cpp wgrad_probe.cpp -o $OUT
# convolution probe (production kernels, no instrumentation); X3ABL=n builds bin/conv_probe_x3abl<n> with the f32x3 K-loop ablation n
XABL=${X3ABL:-0}
COUT=bin/conv_probe; [ "$XABL" != "0" ] && COUT=bin/conv_probe_x3abl$XABL
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function -DOSVOS_X3_ABL=$XABL \
  $C/conv3x3_f32.hip $C/conv3x3_f32x3.hip $C/conv3x3_bf16.hip $C/conv3x3_bf16_dma.hip $C/pack.hip -x hip $C/errors.cpp conv_probe.cpp -o $COUT
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 tr_probe.cpp -o bin/tr_probe
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 mfma_probe.cpp -o bin/mfma_probe
# round-6 micro-benchmarks (single-file HIP programs)
for p in mfma_valu_overlap valu_rate mfma_vs_other mfma_f16_probe; do /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 $p.hip -o bin/$p; done
