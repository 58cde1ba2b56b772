// What does the bf16 matrix pipe of THIS chip sustain?  Register-only loop of v_mfma_f32_32x32x16_bf16 (no LDS, no memory), operand bits selectable:
// the pipe's power draw -- and with it the clock the chip holds -- depends on how many operand bits toggle.
// Build: tools/native/build.sh    Run: tools/native/bin/mfma_probe [zero|const|noise] [waves per workgroup 4|8] [MFMAs per loop turn]
// Prints TFLOP/s; under `rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES` the clock and the pipe occupancy of the same loop.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(r_)); exit(1); } } while (0)

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(512) void mfma_loop(const u32x4* seed, float* out, int iters) {
  // per-lane operands from memory (so the compiler cannot fold them); NACC independent accumulators
  u32x4 a = seed[threadIdx.x & 63], b = seed[64 + (threadIdx.x & 63)];
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc[i], 0, 0, 0);
    // rotate the operand bits a little so that consecutive MFMAs do not see identical inputs (noise mode keeps toggling; zero stays zero)
    a = u32x4{a.y, a.z, a.w, a.x};
    b = u32x4{b.w, b.x, b.y, b.z};
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) out[0] = s;
}

static uint16_t bf16(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "noise";
  const int waves = argc > 2 ? atoi(argv[2]) : 8;
  uint16_t h[128 * 8];
  uint32_t s = 777u;
  for (auto& v : h) {
    s = s * 1664525u + 1013904223u;
    const float r = ((s >> 8) & 0xffff) / 65536.0f - 0.5f;
    v = bf16(!strcmp(mode, "zero") ? 0.f : !strcmp(mode, "const") ? 0.25f : r);
  }
  u32x4* d; float* o;
  CK(hipMalloc(&d, sizeof(h))); CK(hipMalloc(&o, 4));
  CK(hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice));
  const int blocks = 256 * 8, iters = 4000;
  constexpr int NACC = 8;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, 0));
    for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(64 * waves), 0, 0, d, o, iters);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double fl = 5.0 * blocks * waves * (double)iters * NACC * 2.0 * 32 * 32 * 16;
    printf("bf16 MFMA-only loop, operands %-5s, %d waves / workgroup: %.0f TFLOP/s (%.1f ms for 5 launches)\n", mode, waves, fl / (ms * 1e-3) / 1e12, ms);
  }
  return 0;
}
