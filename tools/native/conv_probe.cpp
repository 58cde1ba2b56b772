// Native (no Python) timing + spot check of the 3x3 convolution kernels: seconds of GPU time per run.
//   conv_probe bf16|f32|x3ps N H W Cin Cout tile[,tile...]      (tile -1 = the library's choice; x3ps: f32x3 tile ids (+100: XCD-local block map) on the PRE-SPLIT pack)
// bf16: bf16 NHWC in, bf16 NHWC out (what the network's bf16 mode runs: osvos_conv3x3_bf16mfma_io); f32: osvos_conv3x3_f32_ws.
// Prints per tile the average of 20 launches and the largest relative error of 256 output samples against a double-precision
// restatement on the host (operands rounded the way the kernel rounds them), so a variant that is fast but wrong shows up here.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../osvos-pytorch_amd/csrc/common.h"
#include "../../osvos-pytorch_amd/csrc/kernels.h"

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(r_)); exit(1); } } while (0)

int main(int argc, char** argv) {
  if (argc < 8) { fprintf(stderr, "usage: conv_probe bf16|f32 N H W Cin Cout tile[,tile...]\n"); return 2; }
  const bool bf = !strcmp(argv[1], "bf16"), x3ps = !strcmp(argv[1], "x3ps");
  const int N = atoi(argv[2]), H = atoi(argv[3]), W = atoi(argv[4]), Cin = atoi(argv[5]), Cout = atoi(argv[6]);
  std::vector<int> tiles;
  for (char* t = strtok(argv[7], ","); t; t = strtok(nullptr, ",")) tiles.push_back(atoi(t));
  const size_t nx = (size_t)N * H * W * Cin, ny = (size_t)N * H * W * Cout, nw = (size_t)Cout * Cin * 9;
  std::vector<float> hx(nx), hw(nw), hb(Cout);
  uint32_t s = 2024u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  // PROBE_DATA=zero | relu (half of the activations zero): the chip's clock follows the operand bits (profiles/r02_mfma_probe.txt)
  const char* pd = getenv("PROBE_DATA");
  const int dm = !pd ? 0 : !strcmp(pd, "zero") ? 1 : !strcmp(pd, "relu") ? 2 : 0;
  for (auto& v : hx) { float r = rnd(); if (dm == 1) r = 0.f; if (dm == 2 && r < 0.f) r = 0.f; v = bf ? bf16_to_f32(f32_to_bf16(r)) : r; }
  for (auto& v : hw) v = rnd() * 0.1f;
  for (auto& v : hb) v = rnd();
  std::vector<uint16_t> hx16;
  if (bf) { hx16.resize(nx); for (size_t i = 0; i < nx; ++i) hx16[i] = f32_to_bf16(hx[i]); }
  void *dx, *dy, *dwp, *dpart = nullptr; float *dw, *db;
  CK(hipMalloc(&dx, nx * (bf ? 2 : 4))); CK(hipMalloc(&dy, ny * (bf ? 2 : 4)));
  CK(hipMemcpy(dx, bf ? (void*)hx16.data() : (void*)hx.data(), nx * (bf ? 2 : 4), hipMemcpyHostToDevice));
  CK(hipMalloc(&dw, nw * 4)); CK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&db, Cout * 4)); CK(hipMemcpy(db, hb.data(), Cout * 4, hipMemcpyHostToDevice));
  const int CinP = bf ? (Cin + 31) / 32 * 32 : (Cin + 7) / 8 * 8, CoutP = osvos_cout_pad(Cout);
  CK(hipMalloc(&dwp, (size_t)9 * CinP * CoutP * 4));
  if ((bf ? osvos_pack_fwd_bf16(dw, dwp, Cout, Cin, 0) : osvos_pack_fwd_f32(dw, (float*)dwp, Cout, Cin, 0))) { fprintf(stderr, "pack: %s\n", osvos_last_error()); return 1; }
  if (!bf) { const size_t pb = osvos_conv3x3_splitk_ws_bytes_f32(N, H, W, Cout); if (pb) CK(hipMalloc(&dpart, pb)); }
  void* dwp3 = nullptr;
  if (x3ps) {
    CK(hipMalloc(&dwp3, osvos_wpack_x3_bytes(Cout, Cin)));
    if (osvos_pack_x3(dw, dwp3, Cout, Cin, 0, 0)) { fprintf(stderr, "pack x3: %s\n", osvos_last_error()); return 1; }
  }
  const int ks_env = getenv("PROBE_KSPLIT") ? atoi(getenv("PROBE_KSPLIT")) : 0;
  auto launch = [&](int tile) {
    if (x3ps) return osvos_conv3x3_f32x3_ps((const float*)dx, nullptr, dwp3, db, nullptr, (float*)dy, N, H, W, Cin, Cout, Cout, 1, tile, ks_env, dpart, 0);
    return bf ? osvos_conv3x3_bf16mfma_io(dx, 1, dwp, db, nullptr, 0, nullptr, dy, N, H, W, Cin, Cout, Cout, 1, tile, 0)
              : osvos_conv3x3_f32_ws((const float*)dx, (const float*)dwp, db, nullptr, (float*)dy, N, H, W, Cin, Cout, Cout, 1, tile, dpart, 0);
  };
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double fl = 2.0 * N * H * W * (double)Cout * Cin * 9;
  std::vector<uint16_t> hy16(bf ? ny : 0); std::vector<float> hy(bf ? 0 : ny);
  // The clock drops within milliseconds of idling (a device-to-host copy of the result is enough) and takes tens of milliseconds
  // to come back: all timing happens first, back to back, warm; the spot checks (copy + host arithmetic) come afterwards.
  for (int i = 0; i < 150; ++i) launch(tiles[0]);
  std::vector<float> us(tiles.size(), -1.f);
  for (int pass = 0; pass < 2; ++pass)
    for (size_t k = 0; k < tiles.size(); ++k) {
      if (launch(tiles[k])) continue;
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 20; ++i) launch(tiles[k]);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      us[k] = ms / 20 * 1e3f;                                // (the second pass overwrites the first)
    }
  for (size_t k = 0; k < tiles.size(); ++k) {
    const int tile = tiles[k];
    CK(hipMemset(dy, 0xff, ny * (bf ? 2 : 4)));
    if (launch(tile)) { printf("tile %3d: refused (%s)\n", tile, osvos_last_error()); continue; }
    CK(hipMemcpy(bf ? (void*)hy16.data() : (void*)hy.data(), dy, ny * (bf ? 2 : 4), hipMemcpyDeviceToHost));
    double worst = 0;
    uint32_t q = 99u;
    for (int j = 0; j < 256; ++j) {
      q = q * 1664525u + 1013904223u;
      const size_t o = (j < 8 ? (j & 1 ? ny - 1 - j : (size_t)j) : (size_t)(q % ny));      // corners first, then random
      const int co = (int)(o % Cout); size_t t = o / Cout;
      const int xx = (int)(t % W); t /= W; const int yy = (int)(t % H); const int n = (int)(t / H);
      double acc = hb[co];
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
        const int iy = yy + r - 1, ix = xx + c - 1;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        const float* px = &hx[(((size_t)n * H + iy) * W + ix) * Cin];
        for (int ci = 0; ci < Cin; ++ci) {
          const float wv = hw[((size_t)co * Cin + ci) * 9 + r * 3 + c];
          acc += (double)px[ci] * (bf ? (double)bf16_to_f32(f32_to_bf16(wv)) : (double)wv);
        }
      }
      if (acc < 0) acc = 0;
      const double got = bf ? (double)bf16_to_f32(hy16[o]) : (double)hy[o];
      const double err = fabs(got - acc) / (fabs(acc) + 1.0);
      if (err > worst) worst = err;
    }
    printf("tile %3d: %8.1f us  %7.1f TFLOP/s   worst sampled error %.2e %s\n", tile, us[k], fl / (us[k] * 1e-6) / 1e12, worst,
           worst > (bf ? 8e-3 : 1e-4) ? "  <-- WRONG" : "");
  }
  return 0;
}
