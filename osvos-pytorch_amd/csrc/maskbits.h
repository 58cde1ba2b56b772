// ReLU masks as ONE BIT per element.
//
// The data gradient of a convolution inside a stage is masked by "the previous layer's activation > 0" (autograd of vgg_osvos.py:41,136-145:
// conv -> ReLU -> conv).  Read from the saved activation that is a second full-size tensor per data gradient, fetched in an epilogue nothing
// overlaps: at batch 12 in the bf16 mode conv1_2's data gradient took 0.78 ms against 0.53 ms without its mask (630 MB of it), 0.47 ms per
// step over the eight masked layers (tools/layer_table.py --nomask).  The forward of a layer whose output is a later mask therefore also writes
// the SIGN BITS of what it stores: [N][H][W][C / 32] 32-bit words, bit b of word g = (activation[.., 32 g + b] > 0) -- 1/16 of the bf16
// tensor, 1/32 of the fp32 one -- and the data gradient reads one word per (pixel, 32-channel block) instead of 32 elements.
//
// Accumulator layout of the convolution epilogues (cout-major MFMA result): lane (li, lh) holds, for one 32-cout block, couts
// 8 q + 4 lh + e (q, e = 0..3) of pixel li -- exactly bit 8 q + 4 lh + e of the block's word; the two half-lanes (l, l + 32) own alternating
// nibbles and are OR-ed with one cross-lane read; the lh = 0 lane stores.
#pragma once
#include "common.h"

typedef unsigned mb_u32x4 __attribute__((ext_vector_type(4)));

__device__ inline unsigned mb_load(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) { return __builtin_amdgcn_raw_buffer_load_b32(rs, byte_off, 0, 0); }
__device__ inline bool mb_test(unsigned word, int q, int lh, int e) { return (word >> (8 * q + 4 * lh + e)) & 1u; }
// this lane's four bits of quad q, still un-shifted by its half (bit 8 q + e)
__device__ inline unsigned mb_bits_f32(const f32x4& v, int q) {
  return ((v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u)) << (8 * q);
}
// four packed bf16 (what the bf16 modes keep): positive <=> the 16-bit pattern is a positive integer
__device__ inline unsigned mb_bits_bf16(const uint2& h, int q) {
  const short a = (short)(h.x & 0xffffu), b = (short)(h.x >> 16), c = (short)(h.y & 0xffffu), d = (short)(h.y >> 16);
  return ((a > 0 ? 1u : 0u) | (b > 0 ? 2u : 0u) | (c > 0 ? 4u : 0u) | (d > 0 ? 8u : 0u)) << (8 * q);
}
// every lane of the wave calls this (uniform control flow); lanes with lh = 1 or an out-of-range offset store nothing
__device__ inline void mb_store(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, unsigned bits, int lh) {
  const unsigned mine = bits << (4 * lh);
  const unsigned full = mine | (unsigned)__shfl_xor((int)mine, 32);
  __builtin_amdgcn_raw_buffer_store_b32(full, rs, lh == 0 ? byte_off : 0x80000000u, 0, 0);
}
