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

// duty-cycle form: ONE wave per SIMD (4 waves, 100 KB of LDS per workgroup keeps it to one workgroup per CU), 8 MFMAs then NOPS x s_nop 15:
// does the clock follow the pipe's occupancy (power feedback) or only the kind of instruction and data?
template <int NOPS>
__global__ __launch_bounds__(256) void mfma_duty(const u32x4* seed, float* out, int iters) {
  extern __shared__ char pad[];
  u32x4 a = seed[threadIdx.x & 63], b = seed[64 + (threadIdx.x & 63)];
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc[i], 0, 0, 0);
    a = u32x4{a.y, a.z, a.w, a.x};
    b = u32x4{b.w, b.x, b.y, b.z};
#pragma unroll
    for (int n = 0; n < NOPS; ++n) asm volatile("s_nop 15");
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) out[0] = s + pad[0];
}

// mix form: two waves per SIMD (8 waves), per 8 MFMAs LDSR x ds_read_b128 (1 KB per wave-instruction, conflict-free, noise data in LDS) and VAL x
// (v_cvt_pk_bf16_f32 + v_pk_add_f32-like work): what do LDS and VALU activity beside the MFMAs cost in sustained matrix throughput?
template <int LDSR, int VAL>
__global__ __launch_bounds__(512) void mfma_mix(const u32x4* seed, float* out, int iters) {
  __shared__ u32x4 lds[512 * 4];
  for (int i = threadIdx.x; i < 512 * 4; i += 512) lds[i] = seed[i & 127];
  __syncthreads();
  u32x4 a = seed[threadIdx.x & 63], b = seed[64 + (threadIdx.x & 63)];
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  u32x4 sink = {0u, 0u, 0u, 0u};
  float v0 = __uint_as_float(a.x & 0x3fffffffu), v1 = __uint_as_float(b.y & 0x3fffffffu), vs = 0.f;
  for (int it = 0; it < iters; ++it) {
    u32x4 rd[LDSR > 0 ? LDSR : 1];
#pragma unroll
    for (int l = 0; l < LDSR; ++l) rd[l] = lds[((it + (l & 3)) & 3) * 512 + ((threadIdx.x + 64 * (l >> 2)) & 511)];
#pragma unroll
    for (int i = 0; i < 8; ++i)
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc[i], 0, 0, 0);
#pragma unroll
    for (int k = 0; k < VAL; ++k) {            // independent-ish VALU work: fma chains on four registers
      v0 = __builtin_fmaf(v0, 1.0001f, v1);
      v1 = __builtin_fmaf(v1, 0.9999f, vs);
      vs = vs + v0;
    }
#pragma unroll
    for (int l = 0; l < LDSR; ++l) { sink.x ^= rd[l].x; sink.y ^= rd[l].w; }
    a = u32x4{a.y, a.z, a.w, a.x};
    b = u32x4{b.w, b.x, b.y, b.z};
  }
  float s = vs + __uint_as_float(sink.x ^ sink.y);
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) out[0] = s;
}

template <int LDSR, int VAL>
static void run_mix(const u32x4* d, float* o, const char* mode) {
  const int blocks = 256 * 8, iters = 4000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, 0));
    for (int k = 0; k < 5; ++k) hipLaunchKernelGGL((mfma_mix<LDSR, VAL>), dim3(blocks), dim3(512), 0, 0, d, o, iters);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double fl = 5.0 * blocks * 8 * (double)iters * 8 * 2.0 * 32 * 32 * 16;
    if (rep == 2) printf("mix form, operands %-5s, two waves per SIMD, per 8 MFMAs %2d ds_read_b128 + %2d x 3 VALU: %.0f TFLOP/s\n", mode, LDSR, VAL, fl / (ms * 1e-3) / 1e12);
  }
}

template <int NOPS>
static void run_duty(const u32x4* d, float* o, const char* mode) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_duty<NOPS>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  const int blocks = 256 * 4, iters = 4000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, 0));
    for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(mfma_duty<NOPS>, dim3(blocks), dim3(256), 100 * 1024, 0, d, o, iters);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double fl = 5.0 * blocks * 4 * (double)iters * 8 * 2.0 * 32 * 32 * 16;
    if (rep == 2) printf("duty form, operands %-5s, one wave per SIMD, %2d x s_nop 15 after every 8 MFMAs: %.0f TFLOP/s\n", mode, NOPS, fl / (ms * 1e-3) / 1e12);
  }
}

static uint16_t bf16(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "noise";
  const int waves = argc > 2 ? atoi(argv[2]) : 8;
  static uint16_t h[128 * 8];
  uint32_t s = 777u;
  for (auto& v : h) {
    s = s * 1664525u + 1013904223u;
    const float r = ((s >> 8) & 0xffff) / 65536.0f - 0.5f;
    v = bf16(!strcmp(mode, "zero") ? 0.f : !strcmp(mode, "const") ? 0.25f : r);
  }
  u32x4* d; float* o;
  CK(hipMalloc(&d, sizeof(h))); CK(hipMalloc(&o, 4));
  CK(hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice));
  if (argc > 3 && !strcmp(argv[3], "duty")) {
    run_duty<0>(d, o, mode); run_duty<4>(d, o, mode); run_duty<8>(d, o, mode); run_duty<12>(d, o, mode); run_duty<20>(d, o, mode);
    return 0;
  }
  if (argc > 3 && !strcmp(argv[3], "mix")) {
    run_mix<0, 0>(d, o, mode); run_mix<4, 0>(d, o, mode); run_mix<8, 0>(d, o, mode); run_mix<16, 0>(d, o, mode);
    run_mix<0, 4>(d, o, mode); run_mix<0, 8>(d, o, mode); run_mix<0, 16>(d, o, mode); run_mix<8, 8>(d, o, mode);
    return 0;
  }
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
