// "P3" tensors: an fp32 NHWC tensor stored as its three bf16 pieces, [N][3][H][W][C] bf16 (plane 0 = hi, 1 = mid, 2 = lo).
//
// v = hi + mid + lo EXACTLY (hi = bf16(v), mid = bf16(v - hi), lo = bf16(v - hi - mid), round-to-nearest-even each time; 8 + 8 + 8
// significand bits, both subtractions exact in fp32): a lossless re-encoding of the fp32 value, 6 bytes instead of 4.  It is the operand
// format of the f32x3 kernels (conv3x3_f32x3.hip: six bf16 MFMA products per fp32 product): with the pieces formed ONCE by the producer's
// epilogue the consuming convolutions and weight gradients stage their operands with LDS-DMA (conv3x3_p3.hip) or plain copies
// (wgrad_p3.hip) -- no conversion, subtraction or register staging is left in any K loop.  Plane 0 alone is the ReLU mask (v > 0 <=> hi > 0).
#pragma once
#include "common.h"

typedef __bf16 p3_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int p3_u32x4 __attribute__((ext_vector_type(4)));

__device__ inline unsigned p3_cvt2(float a, float b) {
  p3_bf16x2 h;
  h[0] = (__bf16)a;
  h[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, h);
}
// two values -> their three piece pairs (low half = a, high half = b)
__device__ inline void p3_split2(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
  p0 = p3_cvt2(a, b);
  float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);
  p1 = p3_cvt2(ra, rb);
  ra -= __uint_as_float(p1 << 16);
  rb -= __uint_as_float(p1 & 0xffff0000u);
  p2 = p3_cvt2(ra, rb);
}
__device__ inline void p3_split4(const f32x4& v, uint2& h, uint2& m, uint2& l) {
  p3_split2(v[0], v[1], h.x, m.x, l.x);
  p3_split2(v[2], v[3], h.y, m.y, l.y);
}
// exact: hi + mid has at most 17 significant bits, (hi + mid) + lo is the original fp32 value
__device__ inline float p3_join(unsigned short h, unsigned short m, unsigned short l) {
  return (__uint_as_float((unsigned)h << 16) + __uint_as_float((unsigned)m << 16)) + __uint_as_float((unsigned)l << 16);
}
// four channels (8 bytes per plane) -> fp32
__device__ inline f32x4 p3_join4(const uint2& h, const uint2& m, const uint2& l) {
  f32x4 v;
  v[0] = p3_join((unsigned short)(h.x & 0xffffu), (unsigned short)(m.x & 0xffffu), (unsigned short)(l.x & 0xffffu));
  v[1] = p3_join((unsigned short)(h.x >> 16), (unsigned short)(m.x >> 16), (unsigned short)(l.x >> 16));
  v[2] = p3_join((unsigned short)(h.y & 0xffffu), (unsigned short)(m.y & 0xffffu), (unsigned short)(l.y & 0xffffu));
  v[3] = p3_join((unsigned short)(h.y >> 16), (unsigned short)(m.y >> 16), (unsigned short)(l.y >> 16));
  return v;
}

// Epilogue helper of the MFMA kernels whose accumulators are cout-major (D = [cout rows][pixel columns]): lane (li, lh) holds, for ITS
// pixel, the couts cobase + 8 q + 4 lh + (0..3) in v[q], q = 0..3 (a 32-cout block).  Lanes li and li + 32 trade quads with
// v_permlane32_swap so that each lane owns 8 CONSECUTIVE couts per half block and the three planes go out as 16-byte stores.
//   rs: buffer resource over the image's three planes; pix2: byte offset of the pixel inside a plane (already multiplied by the channel
//   stride and by 2), or 0x80000000 for a pixel outside the image; plane_bytes: H * W * cs * 2; Cout: channels that exist (multiple of 8)
__device__ inline void p3_store32(const f32x4 (&v)[4], const __amdgpu_buffer_rsrc_t& rs, unsigned pix2, unsigned plane_bytes, int cobase, int lh,
                                  int Cout) {
  uint2 pc[3][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) p3_split4(v[q], pc[0][q], pc[1][q], pc[2][q]);
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int pq = 0; pq < 2; ++pq) {
      const auto sx = __builtin_amdgcn_permlane32_swap(pc[p][2 * pq].x, pc[p][2 * pq + 1].x, false, false);
      const auto sy = __builtin_amdgcn_permlane32_swap(pc[p][2 * pq].y, pc[p][2 * pq + 1].y, false, false);
      const p3_u32x4 o = {sx[0], sy[0], sx[1], sy[1]};
      const int co = cobase + 16 * pq + 8 * lh;
      const unsigned off = (co < Cout && pix2 != 0x80000000u) ? pix2 + (unsigned)p * plane_bytes + (unsigned)co * 2u : 0x80000000u;
      __builtin_amdgcn_raw_buffer_store_b128(o, rs, off, 0, 0);
    }
}
