// Fused epilogues of the f32x3 convolution (conv3x3_f32x3.hip): the forward max-pool of a stage boundary folded into the convolution in
// front of it (reference vgg_osvos.py:140 nn.MaxPool2d(2, 2, ceil_mode=True)) and the one-bit ReLU masks (maskbits.h).
//   * the LAST convolution of a stage also writes pooled = maxpool2x2_ceil(relu(y)) -- a wave holds both rows of every window (two M
//     blocks, or the two 16-lane halves of one), the horizontal neighbour is lane ^ 1
// (The BACKWARD pool in the epilogue of the next data gradient was built and measured in round 3 -- backward 2.817 -> 2.861 ms at batch 1:
//  the fused workgroup's tail is 12 dependent memory instructions per accumulator quad on a CU that holds nothing else -- and removed in
//  round 4; docs/DESIGN_rounds_1-4.md 3.8 keeps the numbers, git history the code.)
#pragma once
#include "common.h"

struct ConvEpi {
  float* pooled = nullptr;             // forward: NHWC [N][ceil(H/2)][ceil(W/2)][Cout], written next to y
  // one-bit ReLU masks (maskbits.h), [N][H][W][Cout / 32] words; dense results with Cout % 32 == 0 only
  const unsigned* mask_bits = nullptr; // data gradient: used instead of the fp32 `mask` by launches that are not cut along K (the finalize kernel of a
                                       // split launch reads `mask`: pass both)
  unsigned* y_bits = nullptr;          // forward: sign bits of the result, written next to y (the launch is then never cut along K by partial-sum launches)
  // stream-K (conv3x3_f32x3.hip): workspace of osvos_conv3x3_f32x3_streamk_ws_bytes(), tickets zeroed once by the caller; NULL = plain grids only
  void* sk_ws = nullptr;
  int sk_grid = 0;                     // 0: automatic (taken when the plain grid would idle CUs); > 0: forced, that many workgroups (tests)
};
