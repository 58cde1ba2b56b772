// Micro-benchmark (gfx950): v_mfma_f32_32x32x16_f16 next to v_mfma_f32_32x32x16_bf16 -- (1) sustained rate of a register-only loop on noise operands
// (the power management sets the clock by how many operand bits toggle: the f16 pieces of the 'fp32h2' split carry 11 significand bits, bf16 pieces 8),
// (2) whether the matrix pipe keeps f16 SUBNORMAL inputs (the middle piece of a two-piece f16 split of a small value is subnormal).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int F16>
__global__ __launch_bounds__(256) void rate(const u4* ops, float* out, int iters) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  u4 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = ops[(t * 8 + i) & 0xffff]; b[i] = ops[(t * 8 + 4 + i) & 0xffff]; }
  f32x16 c[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (F16) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a[j]), __builtin_bit_cast(h8, b[(i + j) & 3]), c[i], 0, 0, 0);
        else c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, a[j]), __builtin_bit_cast(b8, b[(i + j) & 3]), c[i], 0, 0, 0);
      }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
  out[t] = s;
}

// one wave: A[k][i] = av for all, B = bv -> every output = 16 av bv
__global__ void denorm(unsigned short abits, unsigned short bbits, float* out) {
  h8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = __builtin_bit_cast(_Float16, abits); b[e] = __builtin_bit_cast(_Float16, bbits); }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}
__global__ void cvt(float v, unsigned* out) {      // v_cvt_pk_f16_f32 of a value in the f16 subnormal range
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 x = {v, v * 0.5f};
  h2 h = __builtin_convertvector(x, h2);
  out[0] = __builtin_bit_cast(unsigned, h);
}

static unsigned short f2h(float f) { _Float16 h = (_Float16)f; unsigned short u; __builtin_memcpy(&u, &h, 2); return u; }
static unsigned short f2b(float f) { unsigned u; __builtin_memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

int main() {
  const int blocks = 256 * 2, iters = 20000;
  u4* ops; float* out;
  (void)hipMalloc(&ops, 65536 * 16); (void)hipMalloc(&out, blocks * 256 * 4 + 64);
  unsigned short* h = (unsigned short*)malloc(65536 * 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  srand(1);
  for (int kind = 0; kind < 4; ++kind) {
    // 0: bf16 noise (N(0,1) rounded), 1: f16 noise (N(0,1) scaled to ~2^10 like a high piece), 2: f16 "middle pieces" (residuals: uniform exponents lower), 3: zeros
    for (int i = 0; i < 65536 * 8; ++i) {
      float u1 = (rand() + 1.f) / (RAND_MAX + 2.f), u2 = rand() / (float)RAND_MAX;
      float g = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
      h[i] = kind == 0 ? f2b(g) : kind == 1 ? f2h(g * 1024.f) : kind == 2 ? f2h(g * 0.4f) : 0;
    }
    (void)hipMemcpy(ops, h, 65536 * 16, hipMemcpyHostToDevice);
    for (int f16 = 0; f16 < 2; ++f16) {
      if ((kind == 0 && f16) || ((kind == 1 || kind == 2) && !f16)) continue;
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        if (f16) hipLaunchKernelGGL(rate<1>, dim3(blocks), dim3(256), 0, 0, ops, out, iters);
        else hipLaunchKernelGGL(rate<0>, dim3(blocks), dim3(256), 0, 0, ops, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
      }
      const double flops = (double)blocks * 4 * iters * 16 * 2.0 * 32 * 32 * 16;
      printf("%-5s operands %-22s: %.2f ms -> %.0f TFLOP/s\n", f16 ? "f16" : "bf16",
             kind == 0 ? "bf16 noise" : kind == 1 ? "f16 noise (high piece)" : kind == 2 ? "f16 noise (mid piece)" : "zeros", best, flops / best * 1e-9);
    }
  }
  // subnormal inputs: a = 2^-20 (f16 subnormal: 0x0010), b = 2^10 -> 16 * 2^-10 = 2^-6 = 0.015625 if kept, 0 if flushed
  float hv;
  hipLaunchKernelGGL(denorm, dim3(1), dim3(64), 0, 0, (unsigned short)0x0010, f2h(1024.f), out);
  (void)hipMemcpy(&hv, out, 4, hipMemcpyDeviceToHost);
  printf("subnormal A (2^-20) x 2^10, K = 16: %g (kept: 0.015625, flushed: 0)\n", hv);
  hipLaunchKernelGGL(denorm, dim3(1), dim3(64), 0, 0, (unsigned short)0x0001, (unsigned short)0x0001, out);
  (void)hipMemcpy(&hv, out, 4, hipMemcpyDeviceToHost);
  printf("subnormal x subnormal (2^-24 x 2^-24), K = 16: %g (kept: %g)\n", hv, 16.0 * ldexp(1.0, -48));
  unsigned hu;
  hipLaunchKernelGGL(cvt, dim3(1), dim3(1), 0, 0, ldexpf(1.25f, -17), (unsigned*)out);
  (void)hipMemcpy(&hu, out, 4, hipMemcpyDeviceToHost);
  printf("v_cvt_pk_f16_f32(1.25 * 2^-17, 1.25 * 2^-18) = 0x%08x (subnormals kept: 0x005000a0)\n", hu);
  return 0;
}
