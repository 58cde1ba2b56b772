// Micro-benchmark (gfx950): issue rate of VALU instructions per SIMD -- one wave per SIMD vs two, several opcodes, 8 independent chains.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define LOOP8(OP) asm volatile(OP " %0, %0, %8\n\t" OP " %1, %1, %8\n\t" OP " %2, %2, %8\n\t" OP " %3, %3, %8\n\t" OP " %4, %4, %8\n\t" OP " %5, %5, %8\n\t" OP " %6, %6, %8\n\t" OP " %7, %7, %8" \
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(k1))
template <int OPC>
__global__ __launch_bounds__(512, 2) void k(int waves, int iters, unsigned* out) {
  if ((int)(threadIdx.x >> 6) >= waves) return;
  unsigned v0 = threadIdx.x * 77u, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
  const unsigned k1 = 0x00010003u;
  for (int i = 0; i < iters; ++i) {
    if (OPC == 0) { LOOP8("v_pk_add_u16"); LOOP8("v_pk_add_u16"); LOOP8("v_pk_add_u16"); LOOP8("v_pk_add_u16"); }
    if (OPC == 1) { LOOP8("v_add_u32"); LOOP8("v_add_u32"); LOOP8("v_add_u32"); LOOP8("v_add_u32"); }
    if (OPC == 2) { LOOP8("v_and_b32"); LOOP8("v_and_b32"); LOOP8("v_and_b32"); LOOP8("v_and_b32"); }
    if (OPC == 3) { LOOP8("v_add_f32"); LOOP8("v_add_f32"); LOOP8("v_add_f32"); LOOP8("v_add_f32"); }
    if (OPC == 4) { LOOP8("v_pk_max_i16"); LOOP8("v_pk_max_i16"); LOOP8("v_pk_max_i16"); LOOP8("v_pk_max_i16"); }
    if (OPC == 5) { LOOP8("v_cvt_pk_bf16_f32"); LOOP8("v_cvt_pk_bf16_f32"); LOOP8("v_cvt_pk_bf16_f32"); LOOP8("v_cvt_pk_bf16_f32"); }
    if (OPC == 6) { LOOP8("v_perm_b32 %0, %0, %8, %8 ; "); }
  }
  out[blockIdx.x * 512 + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}
template <int OPC>
void run(const char* name, unsigned* out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 5000, blocks = 256;
  for (int waves : {4, 8}) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k<OPC>, dim3(blocks), dim3(512), 0, 0, waves, iters, out);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best) best = ms;
    }
    const double per_simd = (double)iters * 32 * (waves / 4);
    printf("%-20s %d wave(s) per SIMD: %.3f ms -> %.2f cycles per wave64 instruction per SIMD at 1.8 GHz (%.2f at 2.4)\n", name, waves / 4, best, best * 1e-3 * 1.8e9 / per_simd,
           best * 1e-3 * 2.4e9 / per_simd);
  }
}
int main() {
  unsigned* out;
  (void)hipMalloc(&out, 256 * 512 * 4);
  run<0>("v_pk_add_u16", out); run<1>("v_add_u32", out); run<2>("v_and_b32", out); run<3>("v_add_f32", out); run<4>("v_pk_max_i16", out); run<5>("v_cvt_pk_bf16_f32", out);
  return 0;
}
