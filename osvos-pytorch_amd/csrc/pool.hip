// 2x2 stride-2 max-pool with ceil_mode=True, NHWC fp32 (reference vgg_osvos.py:140,
// nn.MaxPool2d(kernel_size=2, stride=2, ceil_mode=True) -> aten::max_pool2d_with_indices) and its
// backward fused with the ReLU backward of the producing conv and the side-branch gradient add.
// HBM-bound glue: one thread per (window, 4-channel quad), 16-byte accesses, no indices stored --
// the backward recomputes the argmax from the saved (post-ReLU) pool input.
#include "common.h"

namespace {

typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
__device__ inline uint2 to_bf16x4(const f32x4& v) {
  bf16x4_t h;
  h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
  return __builtin_bit_cast(uint2, h);
}

// ybf / dxbf (optional): bf16 copy of the result, the operand format of the bf16-MFMA convolutions that consume it
__global__ void maxpool_f32_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ y, uint2* __restrict__ ybf,
                                   int N, int H, int W, int C4) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const long total = (long)N * Ho * Wo * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    long t = i / C4;
    const int ox = (int)(t % Wo);
    t /= Wo;
    const int oy = (int)(t % Ho);
    const long n = t / Ho;
    const int iy = 2 * oy, ix = 2 * ox;
    const bool vx = ix + 1 < W, vy = iy + 1 < H;   // clipped (never padded) partial windows
    const f32x4* p = x + ((n * H + iy) * W + ix) * C4 + c;
    f32x4 m = p[0];
    if (vx) { f32x4 v = p[C4]; for (int k = 0; k < 4; ++k) m[k] = v[k] > m[k] ? v[k] : m[k]; }
    if (vy) {
      f32x4 v = p[(long)W * C4];
      for (int k = 0; k < 4; ++k) m[k] = v[k] > m[k] ? v[k] : m[k];
      if (vx) { f32x4 u = p[(long)W * C4 + C4]; for (int k = 0; k < 4; ++k) m[k] = u[k] > m[k] ? u[k] : m[k]; }
    }
    y[i] = m;
    if (ybf != nullptr) ybf[i] = to_bf16x4(m);
  }
}

// dx[pos] = (x[pos] > 0) * ( (pos == first argmax of the window) * dy + dside[pos] )
__global__ void maxpool_bwd_f32_kernel(const f32x4* __restrict__ x, const f32x4* __restrict__ dy,
                                       const f32x4* __restrict__ dside, f32x4* __restrict__ dx, uint2* __restrict__ dxbf,
                                       int N, int H, int W, int C4) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const long total = (long)N * Ho * Wo * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    long t = i / C4;
    const int ox = (int)(t % Wo);
    t /= Wo;
    const int oy = (int)(t % Ho);
    const long n = t / Ho;
    const int iy = 2 * oy, ix = 2 * ox;
    const bool vx = ix + 1 < W, vy = iy + 1 < H;
    const long o00 = ((n * H + iy) * W + ix) * C4 + c;
    const long off[4] = {o00, o00 + C4, o00 + (long)W * C4, o00 + (long)W * C4 + C4};
    const bool valid[4] = {true, vx, vy, vx && vy};
    f32x4 v[4], s[4];
    for (int q = 0; q < 4; ++q) {
      v[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      s[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (valid[q]) {
        v[q] = x[off[q]];
        if (dside != nullptr) s[q] = dside[off[q]];
      }
    }
    const f32x4 g = dy[i];
    f32x4 out[4];
    for (int k = 0; k < 4; ++k) {
      int bi = 0;
      float best = v[0][k];
      for (int q = 1; q < 4; ++q)          // scan order (0,0) (0,1) (1,0) (1,1); strict > keeps the first max
        if (valid[q] && v[q][k] > best) { best = v[q][k]; bi = q; }
      for (int q = 0; q < 4; ++q) {
        const float gq = (q == bi ? g[k] : 0.f) + s[q][k];
        out[q][k] = v[q][k] > 0.f ? gq : 0.f;
      }
    }
    for (int q = 0; q < 4; ++q)
      if (valid[q]) {
        dx[off[q]] = out[q];
        if (dxbf != nullptr) dxbf[off[q]] = to_bf16x4(out[q]);
      }
  }
}

// ---- bf16 tensors (trunk activations / gradients of the bf16-store mode): one thread per (window, 8-channel group) ----
__device__ inline float bf2f(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
__device__ inline unsigned short f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
struct U8 { unsigned short v[8]; };
static_assert(sizeof(U8) == 16, "8 bf16 = 16 bytes");

__global__ void maxpool_bf16_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int N, int H, int W, int C8) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
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
    const uint4* p = x + ((n * H + iy) * W + ix) * C8 + c;
    U8 m = __builtin_bit_cast(U8, p[0]);
    auto upd = [&](const uint4& q) {
      const U8 v = __builtin_bit_cast(U8, q);
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (bf2f(v.v[k]) > bf2f(m.v[k])) m.v[k] = v.v[k];
    };
    if (vx) upd(p[C8]);
    if (vy) {
      upd(p[(long)W * C8]);
      if (vx) upd(p[(long)W * C8 + C8]);
    }
    y[i] = __builtin_bit_cast(uint4, m);
  }
}

// same rule as maxpool_bwd_f32_kernel, arithmetic in fp32, result rounded to bf16 (RNE)
__global__ void maxpool_bwd_bf16_kernel(const uint4* __restrict__ x, const uint4* __restrict__ dy, const uint4* __restrict__ dside,
                                        uint4* __restrict__ dx, int N, int H, int W, int C8) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
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
    const long o00 = ((n * H + iy) * W + ix) * C8 + c;
    const long off[4] = {o00, o00 + C8, o00 + (long)W * C8, o00 + (long)W * C8 + C8};
    const bool valid[4] = {true, vx, vy, vx && vy};
    U8 v[4], s[4];
    for (int q = 0; q < 4; ++q) {
      v[q] = U8{{0, 0, 0, 0, 0, 0, 0, 0}};
      s[q] = U8{{0, 0, 0, 0, 0, 0, 0, 0}};
      if (valid[q]) {
        v[q] = __builtin_bit_cast(U8, x[off[q]]);
        if (dside != nullptr) s[q] = __builtin_bit_cast(U8, dside[off[q]]);
      }
    }
    const U8 g = __builtin_bit_cast(U8, dy[i]);
    U8 out[4];
    for (int k = 0; k < 8; ++k) {
      int bi = 0;
      float best = bf2f(v[0].v[k]);
      for (int q = 1; q < 4; ++q)
        if (valid[q] && bf2f(v[q].v[k]) > best) { best = bf2f(v[q].v[k]); bi = q; }
      for (int q = 0; q < 4; ++q) {
        const float gq = (q == bi ? bf2f(g.v[k]) : 0.f) + bf2f(s[q].v[k]);
        out[q].v[k] = bf2f(v[q].v[k]) > 0.f ? f2bf(gq) : (unsigned short)0;
      }
    }
    for (int q = 0; q < 4; ++q)
      if (valid[q]) dx[off[q]] = __builtin_bit_cast(uint4, out[q]);
  }
}

inline int grid_for(long total) {
  long b = (total + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace

int osvos_maxpool2x2_f32(const float* x, float* y, void* ybf, int N, int H, int W, int C, hipStream_t stream) {
  OSVOS_ARG_CHECK(x && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "maxpool: bad arguments (C=%d)", C);
  const long total = (long)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
  hipLaunchKernelGGL(maxpool_f32_kernel, dim3(grid_for(total)), dim3(256), 0, stream,
                     reinterpret_cast<const f32x4*>(x), reinterpret_cast<f32x4*>(y), reinterpret_cast<uint2*>(ybf), N, H, W, C / 4);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

int osvos_maxpool2x2_bwd_f32(const float* x, const float* dy, const float* dside, float* dx, void* dxbf,
                             int N, int H, int W, int C, hipStream_t stream) {
  OSVOS_ARG_CHECK(x && dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "maxpool_bwd: bad arguments (C=%d)", C);
  const long total = (long)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
  hipLaunchKernelGGL(maxpool_bwd_f32_kernel, dim3(grid_for(total)), dim3(256), 0, stream,
                     reinterpret_cast<const f32x4*>(x), reinterpret_cast<const f32x4*>(dy),
                     reinterpret_cast<const f32x4*>(dside), reinterpret_cast<f32x4*>(dx), reinterpret_cast<uint2*>(dxbf), N, H, W, C / 4);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

int osvos_maxpool2x2_bf16(const void* x, void* y, int N, int H, int W, int C, hipStream_t stream) {
  OSVOS_ARG_CHECK(x && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "maxpool bf16: bad arguments (C=%d)", C);
  const long total = (long)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8);
  hipLaunchKernelGGL(maxpool_bf16_kernel, dim3(grid_for(total)), dim3(256), 0, stream, reinterpret_cast<const uint4*>(x),
                     reinterpret_cast<uint4*>(y), N, H, W, C / 8);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

int osvos_maxpool2x2_bwd_bf16(const void* x, const void* dy, const void* dside, void* dx, int N, int H, int W, int C, hipStream_t stream) {
  OSVOS_ARG_CHECK(x && dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "maxpool_bwd bf16: bad arguments (C=%d)", C);
  const long total = (long)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8);
  hipLaunchKernelGGL(maxpool_bwd_bf16_kernel, dim3(grid_for(total)), dim3(256), 0, stream, reinterpret_cast<const uint4*>(x),
                     reinterpret_cast<const uint4*>(dy), reinterpret_cast<const uint4*>(dside), reinterpret_cast<uint4*>(dx), N, H, W, C / 8);
  OSVOS_LAUNCH_CHECK();
  return 0;
}
