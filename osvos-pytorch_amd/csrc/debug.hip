// Debug references used only by tests: a one-thread-per-output direct convolution (to separate
// MFMA-tiling bugs from plumbing bugs in GPU test logs) and an MFMA fragment-layout probe.
#include "common.h"

namespace {

__global__ void conv3x3_naive_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                     float* __restrict__ y, int N, int H, int W, int Cin, int Cin_s, int Cout, int relu) {
  const long total = (long)N * H * W * Cout;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int co = (int)(i % Cout);
    long t = i / Cout;
    const int ox = (int)(t % W);
    t /= W;
    const int oy = (int)(t % H);
    const long n = t / H;
    float acc = bias ? bias[co] : 0.f;
    for (int r = 0; r < 3; ++r) {
      const int iy = oy + r - 1;
      if (iy < 0 || iy >= H) continue;
      for (int s = 0; s < 3; ++s) {
        const int ix = ox + s - 1;
        if (ix < 0 || ix >= W) continue;
        const float* xp = x + ((n * H + iy) * W + ix) * Cin_s;
        for (int ci = 0; ci < Cin; ++ci) acc = fmaf(xp[ci], w[((long)co * Cin + ci) * 9 + r * 3 + s], acc);
      }
    }
    y[i] = (relu && acc < 0.f) ? 0.f : acc;
  }
}

// out[v][lane][r], v = 0: only k=0 operands non-zero, v = 1: only k=1, v = 2: both, v = 3: lane/reg echo
__global__ void mfma_layout_kernel(float* out) {
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  for (int v = 0; v < 3; ++v) {
    const bool on = (v == 2) || (lh == v);
    const float av = on ? (float)(li + 1) * (lh ? 3.f : 1.f) : 0.f;          // A[i][k] = (i+1) * (k ? 3 : 1)
    const float bv = on ? (float)(li + 1) * 100.f * (lh ? 7.f : 1.f) : 0.f;  // B[k][j] = 100 (j+1) * (k ? 7 : 1)
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[(v * 64 + lane) * 16 + r] = c[r];
  }
  for (int r = 0; r < 16; ++r) out[(3 * 64 + lane) * 16 + r] = (float)(lane * 16 + r);
}

// MFMA-only loop: 4 independent 32x32 accumulators per wave, no memory traffic; gives the fp32 MFMA
// rate this chip sustains at its actual (power-managed) clock -- the practical ceiling of the conv kernels
__global__ __launch_bounds__(256) void mfma_peak_kernel(float* out, int iters) {
  f32x16 c0, c1, c2, c3;
  for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; c2[r] = 0.f; c3[r] = 0.f; }
  float a = (float)(threadIdx.x & 7) * 0.01f, b = (float)(threadIdx.x & 3) * 0.02f;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// bf16 MFMA-only loop: 8 independent accumulators per wave, operands from memory (so the data is the caller's: the chip clocks
// 2.37 GHz on all-zero operands and 1.81 GHz on noise -- what the bf16 pipe sustains depends on how many operand bits toggle)
__global__ __launch_bounds__(512) void mfma_peak_bf16_kernel(const uint4* __restrict__ seed, float* out, int iters) {
  typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
  uint4 a = seed[threadIdx.x & 63], b = seed[64 + (threadIdx.x & 63)];
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc[i], 0, 0, 0);
    a = uint4{a.y, a.z, a.w, a.x};
    b = uint4{b.w, b.x, b.y, b.z};
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// LDS-DMA probe (buffer_load_dwordx4 ... lds): 4 waves; wave w issues two DMA instructions, lane l fetching the 16-byte group
// src[perm] with perm = (l * 7 + 3 * w + i) % nsrc (out-of-range for l == 5: must land as zeros); the LDS image is copied out.
__global__ __launch_bounds__(256) void lds_dma_probe_kernel(const float* __restrict__ src, int nsrc, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float lds[8 * 64 * 4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nsrc * 16, 0x00020000);
  for (int i = threadIdx.x; i < 8 * 64 * 4; i += 256) lds[i] = -1.f;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const unsigned off = lane == 5 ? 0x80000000u : (unsigned)(((lane * 7 + 3 * w + i) % nsrc) * 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + (2 * w + i) * 256), 16, off, 0, 0, 0);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
  __syncthreads();
  for (int i = threadIdx.x; i < 8 * 64 * 4; i += 256) out[i] = lds[i];
}

}  // namespace

extern "C" int osvos_debug_lds_dma(const float* src, int ngroups, float* out, void* stream) {
  OSVOS_ARG_CHECK(src && out && ngroups > 0, "debug lds dma: bad arguments");
  hipLaunchKernelGGL(lds_dma_probe_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, src, ngroups, out);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

// runs `blocks` workgroups x 4 waves x iters x 4 MFMAs; FLOPs = blocks*4*iters*4*2*32*32*2
extern "C" int osvos_debug_mfma_peak(float* out, int blocks, int iters, void* stream) {
  OSVOS_ARG_CHECK(out && blocks > 0 && iters > 0, "debug mfma peak: bad arguments");
  hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

// runs `blocks` workgroups x 8 waves x iters x 8 MFMAs (32x32x16 bf16); FLOPs = blocks*8*iters*8*2*32*32*16; seed: 128 x 16 bytes of bf16 operands
extern "C" int osvos_debug_mfma_peak_bf16(const void* seed, float* out, int blocks, int iters, void* stream) {
  OSVOS_ARG_CHECK(seed && out && blocks > 0 && iters > 0, "debug mfma peak bf16: bad arguments");
  hipLaunchKernelGGL(mfma_peak_bf16_kernel, dim3(blocks), dim3(512), 0, (hipStream_t)stream, reinterpret_cast<const uint4*>(seed), out, iters);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

extern "C" int osvos_debug_conv3x3_naive(const float* x, const float* w_oihw, const float* bias, float* y,
                                         int N, int H, int W, int Cin, int Cin_s, int Cout, int relu, void* stream) {
  OSVOS_ARG_CHECK(x && w_oihw && y, "debug conv: null pointer");
  const long total = (long)N * H * W * Cout;
  long b = (total + 255) / 256;
  if (b > 65535) b = 65535;
  hipLaunchKernelGGL(conv3x3_naive_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, x, w_oihw, bias, y, N, H, W, Cin, Cin_s, Cout, relu);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

extern "C" int osvos_debug_mfma_layout(float* out, void* stream) {
  OSVOS_ARG_CHECK(out, "debug mfma: null pointer");
  hipLaunchKernelGGL(mfma_layout_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
  OSVOS_LAUNCH_CHECK();
  return 0;
}
