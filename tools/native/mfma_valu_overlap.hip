// Micro-benchmark (gfx950): do VALU instructions of one wave overlap with the MFMAs of ANOTHER wave on the same SIMD?
// 8 waves per workgroup (2 per SIMD: w and w + 4).  Modes: 0 = all 8 waves run an MFMA loop; 1 = waves 0-3 MFMA, waves 4-7 idle (exit);
// 2 = waves 0-3 MFMA, waves 4-7 an independent-VALU loop (v_pk_add_u16 chains, 8 accumulators); 3 = waves 4-7 VALU only (0-3 exit);
// 4 = waves 0-3 MFMA, 4-7 VALU with s_setprio 0 and MFMA waves at s_setprio 3.  Prints time and MFMA rate per mode.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512, 2) void k(int mode, int iters, float* out, int valu_per_iter) {
  const int wave = threadIdx.x >> 6;
  const bool mf = mode == 0 || (wave < 4 && mode != 3);
  const bool va = (mode == 2 || mode == 3 || mode == 4) && wave >= 4;
  if (!mf && !va) return;
  if (mf) {
    if (mode == 4) __builtin_amdgcn_s_setprio(3);
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x * 3 + i)); }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  } else {
    unsigned v0 = threadIdx.x * 77u, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
    const unsigned k1 = 0x00010003u;
    const int groups = valu_per_iter / 8;
    for (int i = 0; i < iters; ++i)       // valu_per_iter VALU instructions per 4 MFMAs of the other wave (8 independent chains)
      for (int j = 0; j < groups; ++j)
        asm volatile("v_pk_add_u16 %0, %0, %8\n\tv_pk_add_u16 %1, %1, %8\n\tv_pk_add_u16 %2, %2, %8\n\tv_pk_add_u16 %3, %3, %8\n\t"
                     "v_pk_add_u16 %4, %4, %8\n\tv_pk_add_u16 %5, %5, %8\n\tv_pk_add_u16 %6, %6, %8\n\tv_pk_add_u16 %7, %7, %8"
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(k1));
    unsigned v[8] = {v0, v1, v2, v3, v4, v5, v6, v7};
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = (float)s;
  }
}

int main(int argc, char** argv) {
  const int blocks = 256, iters = 20000;
  float* out;
  hipMalloc(&out, blocks * 512 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int vpi : {8, 16, 24, 32}) {
    for (int mode = 0; mode < 5; ++mode) {
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, mode, iters, out, vpi);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
      }
      const double mf_waves = mode == 0 ? 8 : (mode == 3 ? 0 : 4);
      const double tf = blocks * mf_waves * iters * 4.0 * 2 * 32 * 32 * 16 / (best * 1e-3) / 1e12;
      printf("VALU per 4 MFMA %2d  mode %d  %.3f ms  MFMA %.0f TFLOP/s  (VALU wave: %.1f cycles per instruction at 1.8 GHz)\n", vpi, mode, best, tf,
             best * 1e-3 * 1.8e9 / ((double)iters * vpi));
    }
  }
  return 0;
}
