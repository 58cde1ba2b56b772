// Internal (non-ABI) launchers implemented by the .hip translation units.
#pragma once
#include "common.h"
#include "epi.h"

int osvos_conv3x3_f32(const float* x, const float* wpk, const float* bias, const float* mask, float* y,
                      int N, int H, int W, int Cin, int Cout, int y_cs, int relu, int tile, hipStream_t stream);
size_t osvos_conv3x3_splitk_ws_bytes_f32(int N, int H, int W, int Cout);
void osvos_conv3x3_force_ksplit(int k);
int osvos_conv3x3_f32_ws(const float* x, const float* wpk, const float* bias, const float* mask, float* y,
                         int N, int H, int W, int Cin, int Cout, int y_cs, int relu, int tile, void* part_ws, hipStream_t stream);
int osvos_conv3x3_splitk_finalize_f32(const float* part, const float* bias, const float* mask, float* y, long npix, int Cout, int y_cs,
                                      int ksplit, int relu, hipStream_t stream);
// f32x3 (conv3x3_f32x3.hip): fp32 tensors and fp32 packs, three-way bf16 split operands on the bf16 matrix pipe
bool osvos_conv3x3_f32x3_applicable(int Cin, int Cout, int y_cs);
int osvos_conv3x3_f32x3_num_tiles(void);
size_t osvos_conv3x3_f32x3_streamk_ws_bytes(void);
size_t osvos_conv3x3_f32x3_streamk_ticket_bytes(void);
int osvos_conv3x3_f32x3(const float* x, const float* wpk, const float* bias, const float* mask, float* y,
                        int N, int H, int W, int Cin, int Cout, int y_cs, int relu, int tile, int ksplit, void* part_ws, hipStream_t stream);
// f32x3 weight gradient (wgrad_f32x3.hip): fp32 x / dy, three-way bf16 split, fp32 slabs + the shared reduce
bool osvos_wgrad_f32x3_applicable(int Cin, int Cin_s, int Cout, int Cout_s);
bool osvos_wgrad_f32x3_skinny_applicable(int Cin, int Cin_s, int Cout, int Cout_s);
size_t osvos_wgrad_f32x3_ws_bytes(int N, int H, int W, int Cin_s, int Cout);
int osvos_conv3x3_wgrad_f32x3(const float* x, const float* dy, void* ws, float* dw, float* db,
                              int N, int H, int W, int Cin, int Cin_s, int Cout, int Cout_s, int accumulate, hipStream_t stream);
size_t osvos_wpack_x3_bytes(int M, int K);
#define OSVOS_PACK_MAX 40
int osvos_pack_bf16_multi(const float* const* ws, void* const* dsts, const int* Couts, const int* Cins, const int* dgrads, int n, hipStream_t stream);
int osvos_pack_x3_multi(const float* const* ws, void* const* dsts, const int* Couts, const int* Cins, const int* dgrads, int n, hipStream_t stream);
int osvos_pack_x3_multi_fmt(const float* const* ws, void* const* dsts, const int* Couts, const int* Cins, const int* dgrads, const int* halfs, int n,
                            hipStream_t stream);
int osvos_pack_x3(const float* w, void* wpk3, int Cout, int Cin, int dgrad, hipStream_t stream);
int osvos_conv3x3_f32x3_ps(const float* x, const float* wpk, const void* wpk3, const float* bias, const float* mask, float* y,
                           int N, int H, int W, int Cin, int Cout, int y_cs, int relu, int tile, int ksplit, void* part_ws, hipStream_t stream);
int osvos_conv3x3_f32x3_epi(const float* x, const float* wpk, const void* wpk3, const float* bias, const float* mask, float* y,
                            int N, int H, int W, int Cin, int Cout, int y_cs, int relu, int tile, int ksplit, void* part_ws, const ConvEpi* epi,
                            hipStream_t stream);
bool osvos_dgrad_c3_applicable(int Cin, int Cout);
int osvos_conv3x3_dgrad_c3_f32(const float* dy, const float* wpk_dgrad, float* dx_nchw, int N, int H, int W, int Cout, hipStream_t stream);
int osvos_conv3x3_dgrad_c3_bf16mfma(const void* dy_bf16, const void* wpk_bf16_dgrad, float* dx_nchw, int N, int H, int W, int Cout, hipStream_t stream);
size_t osvos_wgrad_ws_bytes_f32(int N, int H, int W, int Cin_s, int Cout);
int osvos_conv3x3_wgrad_f32(const float* x, const float* dy, void* ws, float* dw, float* db,
                            int N, int H, int W, int Cin, int Cin_s, int Cout, int Cout_s,
                            int accumulate, hipStream_t stream);
int osvos_nchw_to_nhwc_f32(const float* src, float* dst, void* dstbf, int N, int C, int H, int W, int cpad, hipStream_t stream);
int osvos_nhwc_to_nchw_f32(const float* src, float* dst, int N, int C, int H, int W, int cs, hipStream_t stream);
int osvos_pack_fwd_f32(const float* w, float* wpk, int Cout, int Cin, hipStream_t stream);
int osvos_pack_dgrad_f32(const float* w, float* wpk, int Cout, int Cin, hipStream_t stream);
int osvos_maxpool2x2_f32(const float* x, float* y, void* ybf, int N, int H, int W, int C, hipStream_t stream);
int osvos_maxpool2x2_bwd_f32(const float* x, const float* dy, const float* dside, float* dx, void* dxbf,
                             int N, int H, int W, int C, hipStream_t stream);
int osvos_maxpool2x2_bf16(const void* x, void* y, int N, int H, int W, int C, hipStream_t stream);
int osvos_maxpool2x2_bf16_code(const void* x, void* y, void* code, int N, int H, int W, int C, hipStream_t stream);
int osvos_maxpool2x2_bwd_bf16_code(const void* code, const void* dy, const void* dside, void* dx, int N, int H, int W, int C, hipStream_t stream);
int osvos_maxpool2x2_bwd_bf16(const void* x, const void* dy, const void* dside, void* dx, int N, int H, int W, int C, hipStream_t stream);
int osvos_head_lowres_f32(const float* prep, const float* wd, const float* bd, const float* wf,
                          float* score, float* fpart, int N, int h, int w, hipStream_t stream);
int osvos_head_bwd_f32(const float* prep, const float* dside, const float* dfused, const float* f1, const float* f16,
                       const float* wd, const float* wf, float* dprep, void* dprep_bf16, double* acc, int N, int H, int W, int h, int w,
                       int scale_idx, hipStream_t stream);
int osvos_head_bwd4_f32(const float* const* prep, const float* const* dside, const float* dfused, const float* const* f1, const float* const* f16,
                        const float* const* wd, const float* wf, float* const* dprep, void* const* dprep_bf16, double* const* acc,
                        int N, int H, int W, const int* hs, const int* ws, hipStream_t stream);      // the four scales in one launch
int osvos_head_bwd_blocks(int N, int h, int w, int scale_idx);   // workgroups (= partial rows of 34 doubles) head_bwd launches
int osvos_sum_partials(const float* x, long count, double* part, int* nblocks, hipStream_t stream);
// generic (non-diagonal upscale weights) head: head_generic.hip
int osvos_head_weff(const float* wup, const float* wf16, float* weff, int k, hipStream_t stream);
int osvos_head_upsample_generic(const float* const* score, const float* const* prep, const float* const* f1, const float* const* weff,
                                const float* fuse_bias, float* const* outs, int N, int H, int W, const int* hs, const int* ws, hipStream_t stream);
int osvos_head_bwd_generic(const float* prep, const float* dside, const float* dfused, const float* f1, const float* weff, const float* wd,
                           float* dprep, void* dprep_bf16, double* acc, int N, int H, int W, int h, int w, int scale_idx, hipStream_t stream);
int osvos_head_tapsum(const float* P, int channels, const float* d, double* G, int N, int H, int W, int h, int w, int scale_idx, hipStream_t stream);
int osvos_head_generic_param_grads(const float* wup, const float* wf16, const double* G, float* dwf16, float* dwup, int k, int accumulate, hipStream_t stream);
int osvos_head_dw1(const double* G1, float* dw, int k, int accumulate, hipStream_t stream);

// bf16-operand MFMA variant of the convolution (fp32 tensors): conv3x3_bf16.hip
int osvos_pack_fwd_bf16(const float* w, void* wpk, int Cout, int Cin, hipStream_t stream);
int osvos_pack_dgrad_bf16(const float* w, void* wpk, int Cout, int Cin, hipStream_t stream);
int osvos_conv3x3_bf16mfma(const float* x, const void* wpk, const float* bias, const float* mask, float* y,
                           int N, int H, int W, int Cin, int Cout, int y_cs, int relu, int tile, hipStream_t stream);
int osvos_conv3x3_bf16mfma_num_tiles(void);
// xb = 1: x is bf16 NHWC; ybf (optional): bf16 copy of y
int osvos_conv3x3_bf16mfma_bits(const void* x, int xb, const void* wpk, const float* bias, const void* mask, int mask_bf16, const unsigned* mask_bits,
                                float* y, void* ybf, unsigned* y_bits, void* pooled_bf16, int N, int H, int W, int Cin, int Cout, int y_cs, int relu, int tile,
                                hipStream_t stream, void* pool_code = nullptr);
int osvos_conv3x3_bf16mfma_io(const void* x, int xb, const void* wpk, const float* bias, const void* mask, int mask_bf16, float* y, void* ybf,
                              int N, int H, int W, int Cin, int Cout, int y_cs, int relu, int tile, hipStream_t stream);
int osvos_conv3x3_bf16mfma_xb_tiles(int* tiles, int max);
// LDS-DMA staged variant (bf16 activations only): conv3x3_bf16_dma.hip; reached through tile ids 30..37 (30-33: 256 px x 128 / 64 co with 4 waves, 512 px x 128 / 64 co with 8 waves; 34, 35: persistent forms; 36, 37: resident-filter persistent forms for Cin = 64)
bool osvos_conv3x3_bf16_dma_applicable(int Cin, int Cout, int y_cs);
// Cin = 64, bf16 in / out: persistent, resident filter, deferred + skewed packed epilogue (conv3x3_bf16_p64.hip; tile id 38)
bool osvos_conv3x3_bf16_p64_applicable(int Cin, int Cout, int y_cs, bool has_y_f32, bool has_tensor_mask, bool has_mask_bits, bool has_y_bits, bool has_pool,
                                       int relu);
int osvos_conv3x3_bf16_p64(const void* x, const void* wpk, const float* bias, const unsigned* mask_bits, void* ybf, unsigned* y_bits, void* pooled_bf16,
                           void* pool_code, int N, int H, int W, int Cout, int y_cs, int relu, int map, hipStream_t stream);
int osvos_conv3x3_bf16_dma(const void* x, const void* wpk, const float* bias, const void* mask, int mask_bf16, const unsigned* mask_bits, float* y, void* ybf,
                           unsigned* y_bits, void* pooled_bf16, int N, int H, int W, int Cin, int Cout, int y_cs, int relu, int variant, int map, hipStream_t stream,
                           void* pool_code = nullptr);

// bf16-operand weight gradient (fp32 tensors): wgrad_bf16.hip
bool osvos_wgrad_bf16_applicable(int Cin_s, int Cout);
size_t osvos_wgrad_bf16_ws_bytes(int N, int H, int W, int Cin_s, int Cout);
// bf16-store mode of the network: the trunk tensors are bf16.  xb: x AND dy are bf16 (wide layers); the skinny fp32 kernels take
// their WIDE operand (dy of conv1_1, x of side_prep) as bf16 and the narrow one as fp32
int osvos_conv3x3_wgrad_bf16mfma_io(const void* x, const void* dy, int xb, void* ws, float* dw, float* db,
                                    int N, int H, int W, int Cin, int Cin_s, int Cout, int Cout_s,
                                    int accumulate, hipStream_t stream);
int osvos_conv3x3_wgrad_small_f32(const void* x, const void* dy, int wide_bf16, void* ws, float* dw, float* db,
                                  int N, int H, int W, int Cin, int Cin_s, int Cout, int Cout_s,
                                  int accumulate, hipStream_t stream);
int osvos_conv3x3_wgrad_bf16mfma(const float* x, const float* dy, void* ws, float* dw, float* db,
                                 int N, int H, int W, int Cin, int Cin_s, int Cout, int Cout_s,
                                 int accumulate, hipStream_t stream);
