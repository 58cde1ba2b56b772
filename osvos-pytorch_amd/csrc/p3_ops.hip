// Bandwidth-bound glue of the P3 storage mode of the f32x3 network (p3.h): layout conversions and the pooling pair with P3 results.
//   * fp32 NHWC <-> P3 (network edges, tests)
//   * 2x2/2 ceil-mode max-pool (reference vgg_osvos.py:140): fp32 in -> P3 (and optionally fp32) out -- the pooled tensor is only ever
//     read by the next stage's convolution and weight gradient, both P3 consumers
//   * its backward fused with the ReLU mask and the side-branch add (pool.hip's rule: first maximum in scan order wins): fp32 in ->
//     P3 out, the upstream gradient of the stage's last convolution
// One thread per (pixel | window, 8-channel group): 2 x 16-byte fp32 accesses, one 16-byte store per plane.
#include "common.h"
#include "kernels.h"
#include "p3.h"

namespace {

__device__ inline void store_p3_8(bf16_t* base, size_t elem, size_t plane_elems, const f32x4& a, const f32x4& b) {
  uint2 h0, m0, l0, h1, m1, l1;
  p3_split4(a, h0, m0, l0);
  p3_split4(b, h1, m1, l1);
  uint4* o = reinterpret_cast<uint4*>(base + elem);
  const size_t pl = plane_elems / 8;
  o[0] = uint4{h0.x, h0.y, h1.x, h1.y};
  o[pl] = uint4{m0.x, m0.y, m1.x, m1.y};
  o[2 * pl] = uint4{l0.x, l0.y, l1.x, l1.y};
}

// src fp32 [N][hw][cs] (channels c < C copied, C <= cd) -> dst P3 [N][3][hw][cd], channels >= C zero
__global__ void f32_to_p3_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int N, long hw, int C, int cs, int cd) {
  const int c8n = cd >> 3;
  const long total = (long)N * hw * c8n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c8n) * 8;
    const long np = i / c8n, n = np / hw, pix = np % hw;
    f32x4 v[2];
    if (c + 8 <= C && cs % 4 == 0) {
      const f32x4* p = reinterpret_cast<const f32x4*>(src + (size_t)np * cs + c);
      v[0] = p[0];
      v[1] = p[1];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e >> 2][e & 3] = c + e < C ? src[(size_t)np * cs + c + e] : 0.f;
    }
    store_p3_8(dst, ((size_t)n * 3 * hw + pix) * cd + c, (size_t)hw * cd, v[0], v[1]);
  }
}

__global__ void p3_to_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, int N, long hw, int C) {
  const int c4n = C >> 2;
  const long total = (long)N * hw * c4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    const long np = i / c4n, n = np / hw, pix = np % hw;
    const uint2* p = reinterpret_cast<const uint2*>(src + ((size_t)n * 3 * hw + pix) * C + c);
    const size_t pl = (size_t)hw * C / 4;
    *reinterpret_cast<f32x4*>(dst + (size_t)np * C + c) = p3_join4(p[0], p[pl], p[2 * pl]);
  }
}

// x fp32 [N][H][W][C] -> y3 P3 [N][3][Ho][Wo][C] (and y fp32 when non-NULL)
__global__ void maxpool_p3_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ y, bf16_t* __restrict__ y3, int N, int H, int W, int C8) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, C4 = 2 * C8;
  const long total = (long)N * Ho * Wo * C8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C8);
    long t = i / C8;
    const int ox = (int)(t % Wo);
    t /= Wo;
    const int oy = (int)(t % Ho);
    const long n = t / Ho;
    const int iy = 2 * oy, ix = 2 * ox;
    const bool vx = ix + 1 < W, vy = iy + 1 < H;   // clipped (never padded) partial windows
    const f32x4* p = x + ((n * H + iy) * W + ix) * C4 + 2 * c;
    f32x4 m0 = p[0], m1 = p[1];
    auto upd = [&](const f32x4* q) {
      const f32x4 a = q[0], b = q[1];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        m0[k] = a[k] > m0[k] ? a[k] : m0[k];
        m1[k] = b[k] > m1[k] ? b[k] : m1[k];
      }
    };
    if (vx) upd(p + C4);
    if (vy) {
      upd(p + (long)W * C4);
      if (vx) upd(p + (long)W * C4 + C4);
    }
    const size_t opix = (size_t)oy * Wo + ox;
    if (y != nullptr) {
      f32x4* o = y + ((size_t)n * Ho * Wo + opix) * C4 + 2 * c;
      o[0] = m0;
      o[1] = m1;
    }
    store_p3_8(y3, ((size_t)n * 3 * Ho * Wo + opix) * (8 * C8) + 8 * c, (size_t)Ho * Wo * 8 * C8, m0, m1);
  }
}

// dx3[pos] = (x[pos] > 0) * ( (pos == first argmax of the window) * dy + dside[pos] ), P3 out (and fp32 when dx != NULL)
__global__ void maxpool_bwd_p3_kernel(const f32x4* __restrict__ x, const f32x4* __restrict__ dy, const f32x4* __restrict__ dside,
                                      f32x4* __restrict__ dx, bf16_t* __restrict__ dx3, int N, int H, int W, int C8) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, C4 = 2 * C8;
  const long total = (long)N * Ho * Wo * C8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C8);
    long t = i / C8;
    const int ox = (int)(t % Wo);
    t /= Wo;
    const int oy = (int)(t % Ho);
    const long n = t / Ho;
    const int iy = 2 * oy, ix = 2 * ox;
    const bool vx = ix + 1 < W, vy = iy + 1 < H;
    const long p00 = (long)iy * W + ix;                      // pixel index inside the image
    const long pixq[4] = {p00, p00 + 1, p00 + W, p00 + W + 1};
    const bool valid[4] = {true, vx, vy, vx && vy};
    f32x4 v[4][2], s[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v[q][0] = v[q][1] = s[q][0] = s[q][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (valid[q]) {
        const long o = ((long)n * H * W + pixq[q]) * C4 + 2 * c;
        v[q][0] = x[o];
        v[q][1] = x[o + 1];
        if (dside != nullptr) { s[q][0] = dside[o]; s[q][1] = dside[o + 1]; }
      }
    }
    const f32x4 g[2] = {dy[2 * i], dy[2 * i + 1]};
    f32x4 out[4][2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        int bi = 0;
        float best = v[0][hh][k];
#pragma unroll
        for (int q = 1; q < 4; ++q)          // scan order (0,0) (0,1) (1,0) (1,1); strict > keeps the first max
          if (valid[q] && v[q][hh][k] > best) { best = v[q][hh][k]; bi = q; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float gq = (q == bi ? g[hh][k] : 0.f) + s[q][hh][k];
          out[q][hh][k] = v[q][hh][k] > 0.f ? gq : 0.f;
        }
      }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (valid[q]) {
        if (dx != nullptr) {
          f32x4* o = dx + ((long)n * H * W + pixq[q]) * C4 + 2 * c;
          o[0] = out[q][0];
          o[1] = out[q][1];
        }
        store_p3_8(dx3, ((size_t)n * 3 * H * W + pixq[q]) * (8 * C8) + 8 * c, (size_t)H * W * 8 * C8, out[q][0], out[q][1]);
      }
  }
}

inline int grid_for(long total) {
  long b = (total + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace

int osvos_f32_to_p3(const float* src, void* dst3, int N, int H, int W, int C, int cs, int cd, hipStream_t stream) {
  OSVOS_ARG_CHECK(src && dst3 && N > 0 && H > 0 && W > 0 && C > 0 && cs >= C && cd >= C && cd % 8 == 0, "f32_to_p3: bad arguments (C=%d cs=%d cd=%d)", C, cs, cd);
  const long total = (long)N * H * W * (cd / 8);
  hipLaunchKernelGGL(f32_to_p3_kernel, dim3(grid_for(total)), dim3(256), 0, stream, src, reinterpret_cast<bf16_t*>(dst3), N, (long)H * W, C, cs, cd);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

int osvos_p3_to_f32(const void* src3, float* dst, int N, int H, int W, int C, hipStream_t stream) {
  OSVOS_ARG_CHECK(src3 && dst && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "p3_to_f32: bad arguments (C=%d)", C);
  const long total = (long)N * H * W * (C / 4);
  hipLaunchKernelGGL(p3_to_f32_kernel, dim3(grid_for(total)), dim3(256), 0, stream, reinterpret_cast<const bf16_t*>(src3), dst, N, (long)H * W, C);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

int osvos_maxpool2x2_p3(const float* x, float* y, void* y3, int N, int H, int W, int C, hipStream_t stream) {
  OSVOS_ARG_CHECK(x && y3 && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "maxpool p3: bad arguments (C=%d)", C);
  const long total = (long)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8);
  hipLaunchKernelGGL(maxpool_p3_kernel, dim3(grid_for(total)), dim3(256), 0, stream, reinterpret_cast<const f32x4*>(x), reinterpret_cast<f32x4*>(y),
                     reinterpret_cast<bf16_t*>(y3), N, H, W, C / 8);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

int osvos_maxpool2x2_bwd_p3(const float* x, const float* dy, const float* dside, float* dx, void* dx3, int N, int H, int W, int C, hipStream_t stream) {
  OSVOS_ARG_CHECK(x && dy && dx3 && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "maxpool_bwd p3: bad arguments (C=%d)", C);
  const long total = (long)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8);
  hipLaunchKernelGGL(maxpool_bwd_p3_kernel, dim3(grid_for(total)), dim3(256), 0, stream, reinterpret_cast<const f32x4*>(x),
                     reinterpret_cast<const f32x4*>(dy), reinterpret_cast<const f32x4*>(dside), reinterpret_cast<f32x4*>(dx),
                     reinterpret_cast<bf16_t*>(dx3), N, H, W, C / 8);
  OSVOS_LAUNCH_CHECK();
  return 0;
}
