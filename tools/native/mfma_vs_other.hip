// Micro-benchmark (gfx950): which instruction classes of ONE wave slow the MFMAs of the OTHER wave on the same SIMD?
// waves 0-3: MFMA loop (4 independent 32x32x16 bf16 accumulators); waves 4-7: a loop of instruction class X, 16 per 4 MFMAs of the partner.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__shared__ unsigned lds_buf[4096];
template <int X>
__global__ __launch_bounds__(512, 2) void k(int iters, float* out, unsigned* sink, unsigned long long* cyc) {
  const int wave = threadIdx.x >> 6;
  if (wave < 4) {
    bf16x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x * 3 + i)); }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
    return;
  }
  if (X == 0) return;      // partner idle
  unsigned v0 = threadIdx.x * 77u, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3;
  int s0 = blockIdx.x, s1 = 3, s2 = 5, s3 = 7;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(sink, 0, 0, 0x00020000);      // zero records: stores are dropped
  const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(sink, 0, 1 << 20, 0x00020000);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (X == 1) asm volatile("v_add_u32 %0, %0, %1\n\tv_add_u32 %1, %1, %2\n\tv_add_u32 %2, %2, %3\n\tv_add_u32 %3, %3, %0" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
      if (X == 2) asm volatile("s_add_i32 %0, %0, %1\n\ts_add_i32 %1, %1, %2\n\ts_add_i32 %2, %2, %3\n\ts_add_i32 %3, %3, %0" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
      if (X == 3) {
        auto p = __builtin_amdgcn_permlane32_swap(v0, v1, false, false); v0 = p[0]; v1 = p[1];
        auto q = __builtin_amdgcn_permlane32_swap(v2, v3, false, false); v2 = q[0]; v3 = q[1];
        auto r = __builtin_amdgcn_permlane32_swap(v0, v2, false, false); v0 = r[0]; v2 = r[1];
        auto t = __builtin_amdgcn_permlane32_swap(v1, v3, false, false); v1 = t[0]; v3 = t[1];
      }
      if (X == 4) { v0 = __builtin_amdgcn_ds_bpermute((threadIdx.x ^ 32) * 4, v0); v1 = __builtin_amdgcn_ds_bpermute((threadIdx.x ^ 32) * 4, v1); v2 = __builtin_amdgcn_ds_bpermute((threadIdx.x ^ 32) * 4, v2); v3 = __builtin_amdgcn_ds_bpermute((threadIdx.x ^ 32) * 4, v3); }
      if (X == 5) { __builtin_amdgcn_raw_buffer_store_b128(u32x4{v0, v1, v2, v3}, rs, threadIdx.x * 16, 0, 0); v0 += 1; }      // one dropped 16-byte store per group of 4
      if (X == 6) { __builtin_amdgcn_raw_buffer_store_b128(u32x4{v0, v1, v2, v3}, rs2, (threadIdx.x & 255) * 16 + (i & 63) * 4096, 0, 0); v0 += 1; }      // a real 16-byte store (L2-resident 1 MB window)
      if (X == 7) asm volatile("v_pk_add_u16 %0, %0, %1\n\tv_pk_add_u16 %1, %1, %2\n\tv_pk_add_u16 %2, %2, %3\n\tv_pk_add_u16 %3, %3, %0" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
      if (X == 8) { v0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v0, 0xB1, 0xF, 0xF, true); v1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v1, 0xB1, 0xF, 0xF, true); v2 += v0; v3 += v1; }
      if (X == 9) { v0 = lds_buf[(threadIdx.x + v0) & 4095]; v1 = lds_buf[(threadIdx.x * 4 + v1) & 4095]; v2 += v0; v3 += v1; }      // two ds_read_b32
      if (X == 10) { asm volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0"); }
      if (X == 11) { s0 = __builtin_amdgcn_readfirstlane(v0 + s0); s1 = __builtin_amdgcn_readfirstlane(v1 + s1); v2 += s0; v3 += s1; }
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = (float)(v0 + v1 + v2 + v3 + s0 + s1 + s2 + s3);
}
template <int X>
void run(const char* name, float* out, unsigned* sink, unsigned long long* cyc) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000, blocks = 256;
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<X>, dim3(blocks), dim3(512), 0, 0, iters, out, sink, cyc);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  static unsigned long long h[1024];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double sum = 0;
  for (int i = 0; i < 1024; ++i) sum += (double)h[i];
  printf("partner: %-44s kernel %.3f ms | MFMA waves: %.1f cycles per MFMA (32 = the pipe's rate)\n", name, best, sum / 1024 / (iters * 4.0));
}
int main() {
  float* out; unsigned* sink; unsigned long long* cyc; (void)hipMalloc(&cyc, 1024 * 8);
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&sink, 1 << 20);
  run<0>("idle", out, sink, cyc); run<1>("16 x v_add_u32", out, sink, cyc); run<7>("16 x v_pk_add_u16", out, sink, cyc); run<2>("16 x s_add_i32", out, sink, cyc); run<10>("16 x s_nop", out, sink, cyc);
  run<3>("16 x v_permlane32_swap", out, sink, cyc); run<8>("8 x v_mov_dpp + 8 x v_add", out, sink, cyc); run<4>("16 x ds_bpermute", out, sink, cyc); run<9>("8 x ds_read_b32 + 8 v_add", out, sink, cyc);
  run<11>("8 x v_readfirstlane + adds", out, sink, cyc); run<5>("4 x buffer_store_b128 (dropped: 0 records)", out, sink, cyc); run<6>("4 x buffer_store_b128 (real, L2 window)", out, sink, cyc);
  return 0;
}
