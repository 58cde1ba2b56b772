// Fused epilogues of the f32x3 convolution (conv3x3_f32x3.hip) and of its split-K finalize kernel: the two pooling kernels of a stage
// boundary folded into the convolutions next to them (reference vgg_osvos.py:140 nn.MaxPool2d(2, 2, ceil_mode=True) and its autograd).
//   * forward: the LAST convolution of a stage also writes pooled = maxpool2x2_ceil(relu(y)) -- a wave holds both rows of every window
//     (two M blocks, or the two 16-lane halves of one), the horizontal neighbour is lane ^ 1
//   * backward: the data gradient of the FIRST convolution of a stage (its result lives at the pooled resolution) is routed straight
//     through the pool: instead of y the kernel writes dx = relu_mask(x) * (route(y, first maximum of the window in scan order) + dside)
//     at the resolution of x (pool.hip's maxpool_bwd rule) -- the pooled-resolution gradient never reaches HBM
// Standalone pooling launches queue for wave slots behind the 256-register MFMA workgroups of the concurrent weight-gradient stream
// (43-93 us per launch for 20-60 us of traffic, profiles/r02_step_timeline.txt); in an epilogue they ride on waves that already run.
#pragma once
#include "common.h"

struct ConvEpi {
  const float* pool_x = nullptr;       // backward: the pool's input (post-ReLU activation), NHWC [N][pool_H][pool_W][Cout]
  const float* pool_dside = nullptr;   // backward: side-branch gradient at that resolution, or NULL
  float* pool_dx = nullptr;            // backward: result; non-NULL selects the fused pool backward (y is then not written)
  int pool_H = 0, pool_W = 0;
  float* pooled = nullptr;             // forward: NHWC [N][ceil(H/2)][ceil(W/2)][Cout], written next to y
  // one-bit ReLU masks (maskbits.h), [N][H][W][Cout / 32] words; dense results with Cout % 32 == 0 only
  const unsigned* mask_bits = nullptr; // data gradient: used instead of the fp32 `mask` by launches that are not cut along K (the finalize kernel of a
                                       // split launch reads `mask`: pass both)
  unsigned* y_bits = nullptr;          // forward: sign bits of the result, written next to y (the launch is then never cut along K)
};

typedef unsigned int epi_u32x4 __attribute__((ext_vector_type(4)));

// One (pooled pixel, 4-channel quad).  g: gradient of the pooled value; (oy, ox): pooled pixel; co: first channel; cs: channel stride of
// x / dside / dx; xrs / srs / drs: buffer resources over ONE image of each (srs may have zero range: reads return 0).  `live` = the pooled
// pixel exists and co < Cout.  Offsets of positions outside the image are pushed out of range: loads return 0, stores are dropped.
__device__ inline void epi_pool_bwd_quad(const f32x4& g, const __amdgpu_buffer_rsrc_t& xrs, const __amdgpu_buffer_rsrc_t& srs, const __amdgpu_buffer_rsrc_t& drs,
                                         int oy, int ox, int PH, int PW, int cs, int co, bool live) {
  constexpr unsigned OOB = 0x80000000u;
  const int Y = 2 * oy, X = 2 * ox;
  const bool vx = X + 1 < PW, vy = Y + 1 < PH;
  const unsigned base = (unsigned)(((Y * PW + X) * cs + co) * 4);
  const unsigned off[4] = {live ? base : OOB, (live && vx) ? base + (unsigned)cs * 4u : OOB, (live && vy) ? base + (unsigned)(PW * cs) * 4u : OOB,
                           (live && vx && vy) ? base + (unsigned)((PW + 1) * cs) * 4u : OOB};
  f32x4 v[4], s[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    v[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, off[p], 0, 0));
    s[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srs, off[p], 0, 0));
  }
  f32x4 o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    int bi = 0;
    float best = v[0][e];
#pragma unroll
    for (int p = 1; p < 4; ++p)          // scan order (0,0) (0,1) (1,0) (1,1); strict > keeps the first maximum (positions outside read 0 <= x)
      if (off[p] != OOB && v[p][e] > best) { best = v[p][e]; bi = p; }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const float gq = (p == bi ? g[e] : 0.f) + s[p][e];
      o[p][e] = v[p][e] > 0.f ? gq : 0.f;
    }
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(epi_u32x4, o[p]), drs, off[p], 0, 0);
}
