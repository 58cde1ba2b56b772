// Two-piece fp16 split with a block exponent ("h2"; precision 'fp32h2', round 6) -- shared by conv3x3_f32x3.hip (forward / data gradient + the
// weight packs) and wgrad_f32x3.hip.  The arithmetic being matched is the reference's fp32 convolution (vgg_osvos.py:41,136-145) and its autograd.
//
//   v 2^e = h + m + r,   h = rne_f16(v 2^e),  m = rne_f16(v 2^e - h),  |r| <= max(2^-24 |v 2^e|, 2^-25)
//
// e is chosen from the largest magnitude of the block the operand belongs to (a workgroup's tile so far, or a layer's filter) so that that
// magnitude lands in [2^14, 2^15): nothing overflows fp16 (max 65504), values down to 2^-17 of the block maximum keep a 22..23-bit significand
// (11 + 11 + the sign of m), smaller ones an absolute error of 2^-40 of the block maximum (fp16 subnormals, which gfx950's matrix pipe keeps:
// tools/native/mfma_f16_probe.hip).  A product a b = (ah + am)(bh + bm) is formed as ah bh + ah bm + am bh -- THREE v_mfma_f32_32x32x16_f16 with
// fp32 accumulation (an f16 x f16 product is exact in fp32); the dropped am bm is <= 2^-22 |a b|.  The six-product bf16 form (f32x3) is short of the
// fp32 product by <= 3 x 2^-24; this one by <= ~2^-21 worst case per product, random in sign -- next to the K x 2^-24 rounding of the fp32
// accumulation both share.  Power-of-two scales are exact: results are un-scaled once, in the epilogue (v_ldexp_f32).
#pragma once
#include "common.h"

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

constexpr int kH2NoScale = 127;      // "no data seen yet" (larger than any block exponent)

// block exponent from the bits of the block's largest |value| (sign bit clear): max 2^e in [2^14, 2^15); clamped so that 2^e and 2^-e are normal
__host__ __device__ inline int h2_exp(unsigned amax_bits) {
  unsigned e = (amax_bits & 0x7fffffffu) >> 23;
  e = e < 15u ? 15u : (e > 253u ? 253u : e);
  return 141 - (int)e;
}
__device__ inline float h2_pow2(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }      // e in [-126, 127]

// eight fp32 values -> eight (h, m) pairs, packed as the two 16-byte LDS slots of the piece planes
__device__ inline void h2_split2(float x, float y, float sc, unsigned& q0, unsigned& q1) {
  const f32x2_t v = {x * sc, y * sc};
  const f16x2_t h = __builtin_convertvector(v, f16x2_t);             // v_cvt_pk_f16_f32 (RNE)
  const f32x2_t r = v - __builtin_convertvector(h, f32x2_t);         // exact
  const f16x2_t m = __builtin_convertvector(r, f16x2_t);
  q0 = __builtin_bit_cast(unsigned, h);
  q1 = __builtin_bit_cast(unsigned, m);
}
template <class U4A, class U4B>
__device__ inline void h2_split8(const U4A& lo, const U4A& hi, float sc, U4B& p0, U4B& p1) {
  const f32x4 a = __builtin_bit_cast(f32x4, lo), b = __builtin_bit_cast(f32x4, hi);
  unsigned h0, h1, h2, h3, m0, m1, m2, m3;
  h2_split2(a[0], a[1], sc, h0, m0);
  h2_split2(a[2], a[3], sc, h1, m1);
  h2_split2(b[0], b[1], sc, h2, m2);
  h2_split2(b[2], b[3], sc, h3, m3);
  p0 = U4B{h0, h1, h2, h3};
  p1 = U4B{m0, m1, m2, m3};
}
// largest magnitude (as bits, sign clear) of the eight values
template <class U4A>
__device__ inline unsigned h2_amax8(const U4A& lo, const U4A& hi, unsigned m) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned a = lo[e] & 0x7fffffffu, b = hi[e] & 0x7fffffffu;
    m = m > a ? m : a;
    m = m > b ? m : b;
  }
  return m;
}
// wave-wide maximum of non-negative 31-bit values -> uniform (SGPR) result: four DPP steps inside a 16-lane row, four readlanes
__device__ inline unsigned h2_wave_max(unsigned m) {
  int v = (int)m;
  auto mx = [](int a, int b) { return a > b ? a : b; };
  v = mx(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true));        // quad_perm [1,0,3,2]
  v = mx(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true));        // quad_perm [2,3,0,1]
  v = mx(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true));       // row_half_mirror
  v = mx(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true));       // row_mirror
  const int a = mx(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16));
  const int b = mx(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48));
  return (unsigned)mx(a, b);
}
