// Native (no Python, no torch import: seconds of GPU time per run) timing + phase probe of the bf16 weight-gradient kernel.
// Build: tools/native/build.sh    Run on the GPU box: tools/native/bin/wgrad_probe [N H W Cin Cout]
// Prints the average launch time, a checksum of dW (to compare variants) and -- from the s_memtime marks of the
// -DOSVOS_WGRAD_PROF build -- where a wave's time goes per patch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../osvos-pytorch_amd/csrc/common.h"
#include "../../osvos-pytorch_amd/csrc/kernels.h"

extern "C" void osvos_debug_set_wgrad_prof_bf16(unsigned long long* p);


#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(r_)); exit(1); } } while (0)

static uint16_t bf16(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }

static void run(int N, int H, int W, int Cin, int Cout) {
  const size_t nx = (size_t)N * H * W * Cin, ny = (size_t)N * H * W * Cout;
  std::vector<uint16_t> hx(nx), hy(ny);
  uint32_t s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  // PROBE_DATA=zero | const | relu (half of x zero, like post-ReLU activations); default: uniform noise -- the matrix pipe's power draw, and with
  // it the clock the chip sustains, depends on how many operand bits toggle
  const char* mode = getenv("PROBE_DATA");
  const int dm = !mode ? 0 : !strcmp(mode, "zero") ? 1 : !strcmp(mode, "const") ? 2 : !strcmp(mode, "relu") ? 3 : 0;
  for (auto& v : hx) { const float r = rnd(); v = bf16(dm == 1 ? 0.f : dm == 2 ? 0.25f : dm == 3 ? (r > 0.f ? r : 0.f) : r); }
  for (auto& v : hy) { const float r = rnd() * 0.25f; v = bf16(dm == 1 ? 0.f : dm == 2 ? 0.125f : r); }
  void *dx, *ddy, *ws; float *dw, *db; unsigned long long* prof;
  CK(hipMalloc(&dx, nx * 2)); CK(hipMalloc(&ddy, ny * 2));
  CK(hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(ddy, hy.data(), ny * 2, hipMemcpyHostToDevice));
  const size_t wsb = osvos_wgrad_bf16_ws_bytes(N, H, W, Cin, Cout);
  CK(hipMalloc(&ws, wsb)); CK(hipMalloc(&dw, (size_t)Cout * Cin * 9 * 4)); CK(hipMalloc(&db, Cout * 4));
  const size_t nprof = (size_t)8192 * 12 * 10;
  CK(hipMalloc(&prof, nprof * 8)); CK(hipMemset(prof, 0, nprof * 8));
  osvos_debug_set_wgrad_prof_bf16(prof);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int phase = 1; phase >= 0; --phase) {        // 1: the MFMA kernel alone, 0: kernel + slab reduce
    osvos_wgrad_set_phase(phase);
    for (int i = 0; i < 3; ++i)
      if (osvos_conv3x3_wgrad_bf16mfma_io(dx, ddy, 1, ws, dw, db, N, H, W, Cin, Cin, Cout, Cout, 0, 0)) { fprintf(stderr, "launch failed: %s\n", osvos_last_error()); exit(1); }
    CK(hipDeviceSynchronize());
    const int reps = getenv("PROBE_REPS") ? atoi(getenv("PROBE_REPS")) : 10;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) osvos_conv3x3_wgrad_bf16mfma_io(dx, ddy, 1, ws, dw, db, N, H, W, Cin, Cin, Cout, Cout, 0, 0);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double fl = 2.0 * N * H * W * (double)Cout * Cin * 9;
    printf("%dx%dx%d %d->%d  %s: %.1f us  %.0f TFLOP/s\n", N, H, W, Cin, Cout, phase ? "kernel only" : "kernel + reduce", ms / reps * 1e3, fl / (ms / reps * 1e-3) / 1e12);
  }
  std::vector<float> hw((size_t)Cout * Cin * 9), hb(Cout);
  CK(hipMemcpy(hw.data(), dw, hw.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), db, Cout * 4, hipMemcpyDeviceToHost));
  double sw = 0, aw = 0, sb = 0;
  for (float v : hw) { sw += v; aw += v < 0 ? -v : v; }
  for (float v : hb) sb += v;
  printf("  checksum dW sum %.6e  |dW| %.6e  db sum %.6e  dW[12345] %.6e\n", sw, aw, sb, (double)hw[12345 % hw.size()]);
  std::vector<unsigned long long> hp(nprof);
  CK(hipMemcpy(hp.data(), prof, nprof * 8, hipMemcpyDeviceToHost));
  double tot[8] = {0}, life = 0; long waves = 0;
  for (size_t w = 0; w < nprof / 10; ++w) {
    const unsigned long long* q = &hp[w * 10];
    if (q[9] == 0) continue;
    for (int k = 0; k < 8; ++k) tot[k] += (double)q[k];
    life += (double)(q[9] - q[8]);
    ++waves;
  }
  if (waves) {
    const char* names[8] = {"prologue + first loads", "barrier 1 (wait for the other waves' MFMAs)", "vmcnt(0): global loads of the patch", "transpose + ds_write",
                            "barrier 2", "issue next patch's loads", "MFMA k-loop", "epilogue (slab stores)"};
    printf("  phase shares of a wave's lifetime (%ld waves, mean lifetime %.0f ticks of s_memtime, 100 MHz):\n", waves, life / waves);
    for (int k = 0; k < 8; ++k) printf("    %-48s %5.1f %%\n", names[k], 100.0 * tot[k] / life);
  }
  CK(hipFree(dx)); CK(hipFree(ddy)); CK(hipFree(ws)); CK(hipFree(dw)); CK(hipFree(db)); CK(hipFree(prof));
}

int main(int argc, char** argv) {
  if (argc == 6) { run(atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5])); return 0; }
  run(12, 120, 214, 256, 256);
  run(12, 60, 107, 512, 512);
  run(12, 240, 427, 128, 128);
  run(12, 480, 854, 64, 64);
  return 0;
}
