// extern "C" surface of libosvos_hip.so: argument checks + dtype dispatch for the per-op entry
// points declared in include/osvos_hip.h.  (The whole-network calls live in net.cpp.)
#include "kernels.h"

// fp32 tensors in HBM for both built dtypes (OSVOS_F32 and OSVOS_F32_BF16MFMA); only the conv / pack entry
// points behave differently for the latter
#define NEED_F32(dtype, what)                                                                     \
  OSVOS_ARG_CHECK(osvos_dtype_built(dtype), "%s: dtype %d not built (fp32 tensors only)", what, (int)(dtype))

extern "C" {

int osvos_nchw_to_nhwc(const float* src, void* dst, int N, int C, int H, int W, int cpad, int dtype, void* stream) {
  NEED_F32(dtype, "nchw_to_nhwc");
  return osvos_nchw_to_nhwc_f32(src, (float*)dst, nullptr, N, C, H, W, cpad, (hipStream_t)stream);
}
int osvos_nchw_to_nhwc_bf16copy(const float* src, void* dst, void* dst_bf16, int N, int C, int H, int W, int cpad, void* stream) {
  return osvos_nchw_to_nhwc_f32(src, (float*)dst, dst_bf16, N, C, H, W, cpad, (hipStream_t)stream);
}
int osvos_nhwc_to_nchw(const void* src, float* dst, int N, int C, int H, int W, int cs, int dtype, void* stream) {
  NEED_F32(dtype, "nhwc_to_nchw");
  return osvos_nhwc_to_nchw_f32((const float*)src, dst, N, C, H, W, cs, (hipStream_t)stream);
}

size_t osvos_wpack_bytes(int Cout, int Cin, int dtype) {
  if (dtype == OSVOS_F32_BF16MFMA) return (size_t)9 * ((Cin + 31) / 32 * 32) * osvos_cout_pad(Cout) * 2;
  return (size_t)9 * osvos_cin_pad(Cin, dtype) * osvos_cout_pad(Cout) * osvos_elem(dtype);
}
size_t osvos_wpack_dgrad_bytes(int Cout, int Cin, int dtype) {
  if (dtype == OSVOS_F32_BF16MFMA) return (size_t)9 * ((Cout + 31) / 32 * 32) * osvos_cout_pad(Cin) * 2;
  return (size_t)9 * osvos_cin_pad(Cout, dtype) * osvos_cout_pad(Cin) * osvos_elem(dtype);
}
int osvos_pack_conv3x3_fwd(const float* w, void* wpk, int Cout, int Cin, int dtype, void* stream) {
  NEED_F32(dtype, "pack_conv3x3_fwd");
  if (dtype == OSVOS_F32_BF16MFMA) return osvos_pack_fwd_bf16(w, wpk, Cout, Cin, (hipStream_t)stream);
  return osvos_pack_fwd_f32(w, (float*)wpk, Cout, Cin, (hipStream_t)stream);
}
int osvos_pack_conv3x3_dgrad(const float* w, void* wpk, int Cout, int Cin, int dtype, void* stream) {
  NEED_F32(dtype, "pack_conv3x3_dgrad");
  if (dtype == OSVOS_F32_BF16MFMA) return osvos_pack_dgrad_bf16(w, wpk, Cout, Cin, (hipStream_t)stream);
  return osvos_pack_dgrad_f32(w, (float*)wpk, Cout, Cin, (hipStream_t)stream);
}

int osvos_conv3x3(const void* x, const void* wpk, const float* bias, const void* mask, void* y,
                  int N, int H, int W, int Cin, int Cout, int y_cs, int relu, int dtype, int tile, void* stream) {
  NEED_F32(dtype, "conv3x3");
  if (dtype == OSVOS_F32_BF16MFMA)
    return osvos_conv3x3_bf16mfma((const float*)x, wpk, bias, (const float*)mask, (float*)y, N, H, W, Cin, Cout, y_cs, relu, tile,
                                  (hipStream_t)stream);
  if (dtype == OSVOS_F32_X3 && tile < 0 && osvos_conv3x3_f32x3_applicable(Cin, Cout, y_cs))
    return osvos_conv3x3_f32x3((const float*)x, (const float*)wpk, bias, (const float*)mask, (float*)y, N, H, W, Cin, Cout, y_cs, relu, -1, 0,
                               nullptr, (hipStream_t)stream);
  return osvos_conv3x3_f32((const float*)x, (const float*)wpk, bias, (const float*)mask, (float*)y,
                           N, H, W, Cin, Cout, y_cs, relu, tile, (hipStream_t)stream);
}

int osvos_conv3x3_f32x3_tiles(void) { return osvos_conv3x3_f32x3_num_tiles(); }
size_t osvos_wpack_x3_bytes_abi(int Cout, int Cin, int dgrad) { return dgrad ? osvos_wpack_x3_bytes(Cin, Cout) : osvos_wpack_x3_bytes(Cout, Cin); }
int osvos_pack_conv3x3_x3(const float* w, void* wpk3, int Cout, int Cin, int dgrad, void* stream) {
  return osvos_pack_x3(w, wpk3, Cout, Cin, dgrad, (hipStream_t)stream);
}
int osvos_conv3x3_x3(const void* x, const void* wpk3, const float* bias, const void* mask, void* y,
                     int N, int H, int W, int Cin, int Cout, int y_cs, int relu, int tile, void* stream) {
  OSVOS_ARG_CHECK(wpk3 != nullptr, "conv3x3_x3: null pack");
  return osvos_conv3x3_f32x3_ps((const float*)x, nullptr, wpk3, bias, (const float*)mask, (float*)y, N, H, W, Cin, Cout, y_cs, relu,
                                tile >= 200 ? tile - 200 : tile, 0, nullptr, (hipStream_t)stream);
}

// stream-K form of osvos_conv3x3_x3 (conv3x3_f32x3.hip): sk_ws of osvos_conv3x3_x3_streamk_ws_bytes() with its first
// osvos_conv3x3_x3_streamk_ticket_bytes() bytes ZERO before the first use; grid 0 = automatic decision, > 0 = forced with that many workgroups
size_t osvos_conv3x3_x3_streamk_ws_bytes(void) { return osvos_conv3x3_f32x3_streamk_ws_bytes(); }
size_t osvos_conv3x3_x3_streamk_ticket_bytes(void) { return osvos_conv3x3_f32x3_streamk_ticket_bytes(); }
int osvos_conv3x3_x3_streamk(const void* x, const void* wpk3, const float* bias, const void* mask, void* y, void* pooled, int N, int H, int W, int Cin,
                             int Cout, int y_cs, int relu, int tile, int grid, void* sk_ws, void* stream) {
  OSVOS_ARG_CHECK(wpk3 != nullptr && sk_ws != nullptr && grid >= 0, "conv3x3_x3_streamk: null pack / workspace");
  ConvEpi epi;
  epi.sk_ws = sk_ws;
  epi.sk_grid = grid;
  epi.pooled = reinterpret_cast<float*>(pooled);
  return osvos_conv3x3_f32x3_epi((const float*)x, nullptr, wpk3, bias, (const float*)mask, (float*)y, N, H, W, Cin, Cout, y_cs, relu,
                                 tile >= 200 ? tile - 200 : tile, 1, nullptr, &epi, (hipStream_t)stream);
}

// bf16-MFMA convolution with explicit operand / result formats: x fp32 (x_is_bf16 = 0) or bf16 NHWC; y fp32 and,
// when y_bf16 != NULL, a bf16 copy of y with the same channel stride (the operand of the next convolution)
int osvos_conv3x3_bf16io(const void* x, int x_is_bf16, const void* wpk, const float* bias, const void* mask, int mask_is_bf16, float* y,
                         void* y_bf16, int N, int H, int W, int Cin, int Cout, int y_cs, int relu, int tile, void* stream) {
  return osvos_conv3x3_bf16mfma_io(x, x_is_bf16 ? 1 : 0, wpk, bias, mask, mask_is_bf16, y, y_bf16, N, H, W, Cin, Cout, y_cs, relu, tile,
                                   (hipStream_t)stream);
}
// weight gradient of the wide layers (Cin_s, Cout multiples of 64) from bf16 x AND dy; dw/db fp32 as osvos_conv3x3_wgrad
int osvos_conv3x3_wgrad_bf16act(const void* x_bf16, const void* dy_bf16, void* ws, float* dw, float* db, int N, int H, int W, int Cin, int Cin_s,
                                int Cout, int Cout_s, int accumulate, void* stream) {
  return osvos_conv3x3_wgrad_bf16mfma_io(x_bf16, dy_bf16, 1, ws, dw, db, N, H, W, Cin, Cin_s, Cout, Cout_s, accumulate, (hipStream_t)stream);
}
int osvos_maxpool2x2_bf16act(const void* x_bf16, void* y_bf16, int N, int H, int W, int C, void* stream) {
  return osvos_maxpool2x2_bf16(x_bf16, y_bf16, N, H, W, C, (hipStream_t)stream);
}
int osvos_maxpool2x2_bwd_bf16act(const void* x_bf16, const void* dy_bf16, const void* dside_bf16, void* dx_bf16, int N, int H, int W, int C,
                                 void* stream) {
  return osvos_maxpool2x2_bwd_bf16(x_bf16, dy_bf16, dside_bf16, dx_bf16, N, H, W, C, (hipStream_t)stream);
}
int osvos_maxpool2x2_bf16act_code(const void* x_bf16, void* y_bf16, void* code, int N, int H, int W, int C, void* stream) {
  return osvos_maxpool2x2_bf16_code(x_bf16, y_bf16, code, N, H, W, C, (hipStream_t)stream);
}
int osvos_maxpool2x2_bwd_bf16act_code(const void* code, const void* dy_bf16, const void* dside_bf16, void* dx_bf16, int N, int H, int W, int C,
                                      void* stream) {
  return osvos_maxpool2x2_bwd_bf16_code(code, dy_bf16, dside_bf16, dx_bf16, N, H, W, C, (hipStream_t)stream);
}
int osvos_conv3x3_bf16act_fused(const void* x_bf16, const void* wpk, const float* bias, const void* mask_bits, void* y_bf16, void* y_bits,
                                void* pooled_bf16, void* pool_code, int N, int H, int W, int Cin, int Cout, int relu, int tile, void* stream) {
  OSVOS_ARG_CHECK(y_bf16 != nullptr, "conv3x3_bf16act_fused: y_bf16 is required");
  return osvos_conv3x3_bf16mfma_bits(x_bf16, 1, wpk, bias, nullptr, 0, reinterpret_cast<const unsigned*>(mask_bits), nullptr, y_bf16,
                                     reinterpret_cast<unsigned*>(y_bits), pooled_bf16, N, H, W, Cin, Cout, Cout, relu, tile, (hipStream_t)stream, pool_code);
}
int osvos_conv3x3_bf16io_tiles(int* tiles, int max) { return osvos_conv3x3_bf16mfma_xb_tiles(tiles, max); }

size_t osvos_conv3x3_splitk_ws_bytes(int N, int H, int W, int Cout, int dtype) {
  (void)dtype;
  return osvos_conv3x3_splitk_ws_bytes_f32(N, H, W, Cout);
}
int osvos_conv3x3_splitk(const void* x, const void* wpk, const float* bias, const void* mask, void* y,
                         int N, int H, int W, int Cin, int Cout, int y_cs, int relu, int dtype, int tile, int ksplit,
                         void* part_ws, void* stream) {
  OSVOS_ARG_CHECK(dtype == OSVOS_F32 || dtype == OSVOS_F32_X3, "conv3x3_splitk: fp32 only (dtype %d)", dtype);
  OSVOS_ARG_CHECK(part_ws != nullptr && ksplit >= 0 && ksplit <= 8, "conv3x3_splitk: bad ksplit / workspace");
  osvos_conv3x3_force_ksplit(ksplit);
  const int rc = osvos_conv3x3_f32_ws((const float*)x, (const float*)wpk, bias, (const float*)mask, (float*)y, N, H, W, Cin, Cout,
                                      y_cs, relu, (dtype == OSVOS_F32_X3 && tile < 0) ? -2 : tile, part_ws, (hipStream_t)stream);
  osvos_conv3x3_force_ksplit(0);
  return rc;
}

size_t osvos_wgrad_ws_bytes(int N, int H, int W, int Cin, int Cout, int dtype) {
  const size_t f = osvos_wgrad_ws_bytes_f32(N, H, W, Cin, Cout);
  const size_t b = dtype == OSVOS_F32_BF16MFMA ? osvos_wgrad_bf16_ws_bytes(N, H, W, Cin, Cout) : osvos_wgrad_f32x3_ws_bytes(N, H, W, Cin, Cout);
  return f > b ? f : b;
}
int osvos_conv3x3_wgrad(const void* x, const void* dy, void* ws, float* dw, float* db,
                        int N, int H, int W, int Cin, int Cin_s, int Cout, int Cout_s,
                        int accumulate, int dtype, void* stream) {
  NEED_F32(dtype, "conv3x3_wgrad");
  // the wide trunk layers go through the bf16 MFMA kernel; conv1_1 (Cin 3) and side_prep (Cout 16) keep
  // their exact-fp32 skinny kernels (5 % of the weight-gradient FLOPs)
  if (dtype == OSVOS_F32_BF16MFMA && Cin == Cin_s && Cout % 64 == 0 && osvos_wgrad_bf16_applicable(Cin_s, Cout))
    return osvos_conv3x3_wgrad_bf16mfma((const float*)x, (const float*)dy, ws, dw, db, N, H, W, Cin, Cin_s, Cout, Cout_s,
                                        accumulate, (hipStream_t)stream);
  // f32x3 (dtype OSVOS_F32_X3): the wide trunk layers on the bf16 matrix pipe with three-way split operands; conv1_1 and side_prep keep
  // their exact skinny kernels
  if (dtype == OSVOS_F32_X3 &&
      (osvos_wgrad_f32x3_applicable(Cin, Cin_s, Cout, Cout_s) || osvos_wgrad_f32x3_skinny_applicable(Cin, Cin_s, Cout, Cout_s)))
    return osvos_conv3x3_wgrad_f32x3((const float*)x, (const float*)dy, ws, dw, db, N, H, W, Cin, Cin_s, Cout, Cout_s, accumulate,
                                     (hipStream_t)stream);
  return osvos_conv3x3_wgrad_f32((const float*)x, (const float*)dy, ws, dw, db, N, H, W, Cin, Cin_s, Cout, Cout_s,
                                 accumulate, (hipStream_t)stream);
}

int osvos_maxpool2x2(const void* x, void* y, int N, int H, int W, int C, int dtype, void* stream) {
  NEED_F32(dtype, "maxpool2x2");
  return osvos_maxpool2x2_f32((const float*)x, (float*)y, nullptr, N, H, W, C, (hipStream_t)stream);
}
int osvos_maxpool2x2_bf16copy(const float* x, float* y, void* y_bf16, int N, int H, int W, int C, void* stream) {
  return osvos_maxpool2x2_f32(x, y, y_bf16, N, H, W, C, (hipStream_t)stream);
}
int osvos_maxpool2x2_bwd(const void* x, const void* dy, const void* dside, void* dx,
                         int N, int H, int W, int C, int dtype, void* stream) {
  NEED_F32(dtype, "maxpool2x2_bwd");
  return osvos_maxpool2x2_bwd_f32((const float*)x, (const float*)dy, (const float*)dside, (float*)dx, nullptr, N, H, W, C, (hipStream_t)stream);
}
int osvos_maxpool2x2_bwd_bf16copy(const float* x, const float* dy, const float* dside, float* dx, void* dx_bf16,
                                  int N, int H, int W, int C, void* stream) {
  return osvos_maxpool2x2_bwd_f32(x, dy, dside, dx, dx_bf16, N, H, W, C, (hipStream_t)stream);
}

int osvos_conv3x3_dgrad_c3(const float* dy, const float* wpk_dgrad, float* dx_nchw, int N, int H, int W, int Cout, void* stream) {
  return osvos_conv3x3_dgrad_c3_f32(dy, wpk_dgrad, dx_nchw, N, H, W, Cout, (hipStream_t)stream);
}
int osvos_conv3x3_dgrad_c3_bf16mma(const void* dy_bf16, const void* wpk_bf16_dgrad, float* dx_nchw, int N, int H, int W, int Cout, void* stream) {
  return osvos_conv3x3_dgrad_c3_bf16mfma(dy_bf16, wpk_bf16_dgrad, dx_nchw, N, H, W, Cout, (hipStream_t)stream);
}


int osvos_head_lowres(const void* prep, const float* wd, const float* bd, const float* wf,
                      float* score, float* fpart, int N, int h, int w, int dtype, void* stream) {
  NEED_F32(dtype, "head_lowres");
  return osvos_head_lowres_f32((const float*)prep, wd, bd, wf, score, fpart, N, h, w, (hipStream_t)stream);
}
int osvos_head_bwd(const void* prep, const float* dside, const float* dfused,
                   const float* f1, const float* f16, const float* wd, const float* wf,
                   void* dprep, double* acc, int N, int H, int W, int h, int w, int scale_idx,
                   int dtype, void* stream) {
  NEED_F32(dtype, "head_bwd");
  return osvos_head_bwd_f32((const float*)prep, dside, dfused, f1, f16, wd, wf, (float*)dprep, nullptr, acc, N, H, W, h, w,
                            scale_idx, (hipStream_t)stream);
}

}  // extern "C"
