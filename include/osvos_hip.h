/* osvos_hip.h -- C ABI of libosvos_hip.so: the MI355X (gfx950) OSVOS forward/backward hot path.
 *
 * The reference (kmaninis/OSVOS-PyTorch) has no FFI boundary of its own: its "operator API" is
 * torch.nn.Module + autograd and every arithmetic op is delegated to ATen.  This header is the
 * boundary a maintainer binds instead (INTEGRATION.md shows the ctypes stub).  Each entry point
 * cites the reference call site whose ATen op(s) it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless it says "host"; the caller (PyTorch) owns all
 *     memory, including workspaces; the library never allocates, frees or synchronises
 *   - every function enqueues on `stream` (a hipStream_t passed as void*) and returns
 *     0 on success, >0 a hipError_t, <0 an argument error; osvos_last_error() has the text
 *   - activations are NHWC ("channels last") float32 (dtype 0) or bfloat16 (dtype 1) unless
 *     a parameter says nchw; parameters arrive in the reference's state_dict layout (OIHW fp32)
 *   - re-entrant: no mutable global state except the per-thread error string
 */
#ifndef OSVOS_HIP_H
#define OSVOS_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OSVOS_ABI_VERSION 1
#define OSVOS_F32 0
#define OSVOS_BF16 1          /* bf16 tensors in HBM: reserved, not built */
#define OSVOS_F32_BF16MFMA 2  /* fp32 tensors in HBM; conv forward/data-gradient operands rounded to bf16 (RNE) while
                                 staged into LDS, v_mfma_f32_32x32x16_bf16 with fp32 accumulate; everything else fp32 */
#define OSVOS_FLAG_GENERIC_DECONV 0x100  /* OR-ed into the dtype of osvos_net_pack / osvos_net_forward / osvos_net_backward (and the *_bytes queries):
                                            upscale[i].weight is NOT diagonal with one shared filter -> the generic transposed-convolution head
                                            (head_generic.hip) runs instead of the commuted one, and gradients of upscale / upscale_ weights
                                            are written when their grads[] entries (0..7) are non-NULL */
#define OSVOS_FLAG_DEFER_JOIN 0x200      /* OR-ed into the dtype of osvos_net_backward: do NOT make `stream` wait for the weight-gradient work still
                                            running on aux_stream / aux2_stream when the call returns (the data-gradient chain, dx and everything the
                                            caller's next forward needs ARE complete in stream order).  The caller owes one osvos_net_join before it reads
                                            a parameter gradient on `stream` (optimizer step, all-reduce) and must keep `ws` alive until then.  Lets the
                                            tail of the weight gradients run under the next micro-batch's forward (gradient-accumulation loops). */
#define OSVOS_FLAG_INFERENCE 0x400       /* OR-ed into the dtype of osvos_net_forward: NO backward will follow (train_online.py:172-181, torch.no_grad): the
                                            forward skips what only a backward reads -- the one-bit ReLU masks and the pool-code bytes -- so the
                                            convolutions that write them keep their fused pooling epilogue and nothing is stored for nobody (round 6;
                                            the workspace may be the inference-sized one, osvos_net_ws_bytes_infer).  Logits are bit-identical. */
#define OSVOS_FLAG_X3_TWO_PIECES 0x800   /* OR-ed into the dtype OSVOS_F32_X3 of osvos_net_forward / osvos_net_backward: precision 'fp32x2' -- the f32x3
                                            kernels take TWO bf16 pieces per operand and form three products (ah*bh + ah*bm + am*bh) instead of three
                                            pieces / six products: operands carry 16 significand bits (TF32, cuDNN's default for fp32 convolutions on
                                            the reference's GPUs, carries 11), accumulation is fp32, half the matrix work.  Tensors, packs and workspaces
                                            are those of OSVOS_F32_X3.  NOT fp32-grade: see profiles/r06_fp32x2.txt for where it lands. */
#define OSVOS_FLAG_X3_HALF_PIECES 0x1000 /* OR-ed into the dtype OSVOS_F32_X3: precision 'fp32h2' -- the f32x3 kernels take TWO FP16 pieces per operand under
                                            block exponents (csrc/h2split.h): 22-23 significand bits per operand, three products on
                                            v_mfma_f32_32x32x16_f16, fp32 accumulation; measured against float64 the error is the exact fp32 kernels'
                                            (tests/test_gpu_ops.py::test_f32x3_kernels_with_fp16_pairs).  osvos_net_forward / osvos_net_backward: the
                                            call's kernels run in that form.  osvos_net_pack: the FORWARD packs are written in the FP16-pair format
                                            (same buffers; a pack and the calls that read it must agree -- a forward with the flag needs forward packs
                                            with it, a backward with the flag needs data-gradient packs with OSVOS_FLAG_X3_HALF_PIECES_BWD). */
#define OSVOS_FLAG_X3_HALF_PIECES_BWD 0x2000 /* osvos_net_pack only: the DATA-GRADIENT packs are written in the FP16-pair format (for backward calls that
                                            carry OSVOS_FLAG_X3_HALF_PIECES) */
#define OSVOS_F32_X3 3        /* fp32 tensors, fp32 parameters and fp32 weight packs exactly as OSVOS_F32; the wide 3x3 convolutions
                                 (forward, data gradient) run on the bf16 matrix pipe with three-way split operands (six bf16
                                 products per fp32 product, fp32 accumulate): fp32-grade results, see osvos_conv3x3 below */
#define OSVOS_NPARAMS 52 /* tensors of OSVOS.state_dict(), reference order */

int osvos_version(void);
const char* osvos_last_error(void);

/* ---- layout helpers -------------------------------------------------------------------- */
/* NCHW fp32 [N,C,H,W] -> NHWC with `cpad` channels (zero filled), replaces the implicit layout
 * of the tensor handed to OSVOS.forward (vgg_osvos.py:59). */
int osvos_nchw_to_nhwc(const float* src, void* dst, int N, int C, int H, int W, int cpad, int dtype, void* stream);
/* NHWC (channel stride cs) -> NCHW fp32 [N,C,H,W] */
int osvos_nhwc_to_nchw(const void* src, float* dst, int N, int C, int H, int W, int cs, int dtype, void* stream);

/* OIHW fp32 [Cout,Cin,3,3] -> forward pack [9][CinP/G][CoutP][G] (G = 4 fp32 / 8 bf16 channels per
 * 16-byte group; CinP = Cin rounded up to 2G, CoutP = Cout rounded up to 32; zero filled). */
size_t osvos_wpack_bytes(int Cout, int Cin, int dtype);
int osvos_pack_conv3x3_fwd(const float* w_oihw, void* wpk, int Cout, int Cin, int dtype, void* stream);
/* same weights packed for the data-gradient: a 3x3 conv of dY with the 180-degree rotated,
 * channel-transposed filter; pack is [9][CoutP'/G][CinP'][G] with roles swapped. */
size_t osvos_wpack_dgrad_bytes(int Cout, int Cin, int dtype);
int osvos_pack_conv3x3_dgrad(const float* w_oihw, void* wpk, int Cout, int Cin, int dtype, void* stream);

/* ---- 3x3 convolution (replaces aten::convolution for nn.Conv2d(k=3,p=1): vgg_osvos.py:41,142
 *      and, with the dgrad pack, the input-gradient half of aten::convolution_backward) ------
 * y[n,h,w,co] = epi( bias[co] + sum_{r,s,ci} x[n,h+r-1,w+s-1,ci] * W[co,ci,r,s] )
 *   x: NHWC, channel stride Cin (multiple of 2G);  y: NHWC, channel stride y_cs, Cout written
 *   bias: fp32 [Cout] or NULL;  relu != 0: epi = max(.,0)  (vgg_osvos.py:143)
 *   mask: NHWC like y or NULL: epi zeroes the result where mask <= 0 (ReLU backward of the
 *         producer layer fused into this layer's data-gradient: aten::threshold_backward)
 *   tile: -1 = automatic, otherwise a tile-config index, +100 for the XCD-local spatial
 *         block mapping (tuning / tests) */
int osvos_conv3x3(const void* x, const void* wpk, const float* bias, const void* mask, void* y,
                  int N, int H, int W, int Cin, int Cout, int y_cs, int relu, int dtype, int tile, void* stream);
int osvos_conv3x3_num_tiles(void);
/* f32x3: the SAME fp32 convolution (fp32 tensors, fp32 packs, dtype OSVOS_F32) evaluated on the bf16 matrix pipe with
 * three-way split operands -- v = hi + mid + lo exactly (three bf16 pieces), a*b ~ six bf16 products accumulated in fp32,
 * dropped terms <= 2^-24 |a b| each: fp32-grade results at up to 2.67x the fp32-MFMA rate.  Needs Cin % 16 == 0,
 * Cout % 4 == 0 and >= 32, y_cs % 4 == 0; other shapes (conv1_1, side_prep) stay on the exact kernel.
 *   The arithmetic is chosen PER CALL: dtype OSVOS_F32_X3 with tile = -1 takes f32x3 wherever it applies, dtype OSVOS_F32 the exact
 *   kernels; tile 200 + k (k < osvos_conv3x3_f32x3_tiles(), +100 for the XCD-local map) forces an f32x3 tile config.  (Rounds 2-3 also had
 *   a process-wide osvos_set_fp32_conv_mode / OSVOS_FP32_CONV switch: removed in round 4 -- no mutable global state behind the ABI.) */
int osvos_conv3x3_f32x3_tiles(void);
/* f32x3 with PRE-SPLIT weights: the three bf16 piece planes of the filter are formed once (osvos_pack_conv3x3_x3; dgrad != 0 packs the
 * rotated / transposed filter of the data gradient) instead of by every workgroup while staging -- same pieces, bit-identical results,
 * less work between the two barriers of a K chunk.  osvos_net_* do this internally for dtype OSVOS_F32_X3 (OSVOS_X3_PRESPLIT=0: don't).
 * osvos_conv3x3_x3: arguments as osvos_conv3x3 with dtype OSVOS_F32_X3; tile -1 or one of the eight-wave f32x3 tiles (10, 12, 14, 15). */
size_t osvos_wpack_x3_bytes_abi(int Cout, int Cin, int dgrad);
int osvos_pack_conv3x3_x3(const float* w_oihw, void* wpk3, int Cout, int Cin, int dgrad, void* stream);
/* Stream-K form of osvos_conv3x3_x3 (round 4): one persistent workgroup per CU walks an even share of the launch's (tile, 16-channel K chunk)
 * units; tiles whose K range is shared are summed by the last arriver in a fixed order (deterministic; no workgroup ever waits for another).
 * sk_ws: caller-owned workspace of osvos_conv3x3_x3_streamk_ws_bytes(); its first osvos_conv3x3_x3_streamk_ticket_bytes() bytes must be ZERO
 * before the first launch that uses the buffer (every launch leaves them zero); launches sharing one workspace must be stream-ordered.
 * grid: 0 = automatic (stream-K only when a plain grid would leave CUs without a tile), > 0 = forced with that many workgroups (<= 256).
 * pooled (optional): maxpool2x2_ceil(y) written next to y (needs relu, y_cs == Cout).  Tiles 10, 12, 14 (or -1); other tiles run plain. */
size_t osvos_conv3x3_x3_streamk_ws_bytes(void);
size_t osvos_conv3x3_x3_streamk_ticket_bytes(void);
int osvos_conv3x3_x3_streamk(const void* x, const void* wpk3, const float* bias, const void* mask, void* y, void* pooled, int N, int H, int W, int Cin,
                             int Cout, int y_cs, int relu, int tile, int grid, void* sk_ws, void* stream);
int osvos_conv3x3_x3(const void* x, const void* wpk3, const float* bias, const void* mask, void* y,
                     int N, int H, int W, int Cin, int Cout, int y_cs, int relu, int tile, void* stream);
/* same convolution cut into `ksplit` parts along K = 9*Cin (0 = automatic, 1..8): layers too small to balance over
 * 256 CUs (conv4_x, conv5_x at batch 1) get more, shorter workgroups; partial sums go to part_ws
 * (osvos_conv3x3_splitk_ws_bytes) and a second kernel applies bias / ReLU / mask.  fp32 only. */
size_t osvos_conv3x3_splitk_ws_bytes(int N, int H, int W, int Cout, int dtype);
int osvos_conv3x3_splitk(const void* x, const void* wpk, const float* bias, const void* mask, void* y,
                         int N, int H, int W, int Cin, int Cout, int y_cs, int relu, int dtype, int tile, int ksplit,
                         void* part_ws, void* stream);

/* Input gradient of the FIRST convolution (Cin = 3; train_online.py:121 makes the input require grad): dy NHWC fp32 [N][H][W][Cout]
 * (Cout % 16 == 0), wpk_dgrad = osvos_pack_conv3x3_dgrad(w, .., Cout, 3, OSVOS_F32) -> dx_nchw fp32 [N][3][H][W] directly.  A bandwidth
 * kernel (fp32 FMAs, filter through scalar loads) in place of a 32-cout MFMA tile that would waste 10x the matrix work. */
int osvos_conv3x3_dgrad_c3(const float* dy, const float* wpk_dgrad, float* dx_nchw, int N, int H, int W, int Cout, void* stream);
/* the same from a bf16 dy (NHWC bf16: the bf16-store mode) on the matrix pipe, in the arithmetic of the bf16 mode (bf16 dy x bf16 filter, fp32
 * accumulation; the whole 64-channel halo tile goes through LDS once, the filter is the MFMA's row operand): Cout = 64 only; wpk_bf16_dgrad =
 * osvos_pack_conv3x3_dgrad(w, .., 64, 3, OSVOS_F32_BF16MFMA).  What the network's bf16-store mode runs for train_parent.py:136's input gradient. */
int osvos_conv3x3_dgrad_c3_bf16mma(const void* dy_bf16, const void* wpk_bf16_dgrad, float* dx_nchw, int N, int H, int W, int Cout, void* stream);


/* ---- bf16 operand storage for the bf16-MFMA path (dtype OSVOS_F32_BF16MFMA) ------------------------------------------
 * The convolutions of that path round their operands to bf16 anyway; producers can hand the rounded tensor over
 * directly so that the consumer reads half the bytes and skips the conversion (same numbers, RNE either way).
 *   osvos_conv3x3_bf16io: x fp32 (x_is_bf16 = 0) or bf16 NHWC; mask fp32 or (mask_is_bf16) bf16; y fp32 and/or y_bf16
 *     (either may be NULL; y_bf16 has the same channel stride and needs Cout % 8 == 0, y_cs % 8 == 0).  Other
 *     arguments as osvos_conv3x3.  With x_is_bf16 only the tile ids osvos_conv3x3_bf16io_tiles() reports are built.
 *   *_bf16copy: the fp32 kernel plus a bf16 copy of its output (NULL = none).
 *   *_bf16act: pooling on bf16 tensors (C % 8 == 0); the backward adds in fp32 and rounds once (RNE). */
int osvos_conv3x3_bf16io(const void* x, int x_is_bf16, const void* wpk, const float* bias, const void* mask, int mask_is_bf16, float* y,
                         void* y_bf16, int N, int H, int W, int Cin, int Cout, int y_cs, int relu, int tile, void* stream);
int osvos_maxpool2x2_bf16act(const void* x_bf16, void* y_bf16, int N, int H, int W, int C, void* stream);
/* weight gradient of the wide layers (Cin_s and Cout multiples of 64, Cin == Cin_s) from bf16 x AND dy; workspace of
 * osvos_wgrad_ws_bytes(.., OSVOS_F32_BF16MFMA); dw/db fp32 as osvos_conv3x3_wgrad (db = fp32 sums of the bf16 dy) */
int osvos_conv3x3_wgrad_bf16act(const void* x_bf16, const void* dy_bf16, void* ws, float* dw, float* db, int N, int H, int W, int Cin, int Cin_s,
                                int Cout, int Cout_s, int accumulate, void* stream);
int osvos_maxpool2x2_bwd_bf16act(const void* x_bf16, const void* dy_bf16, const void* dside_bf16, void* dx_bf16, int N, int H, int W, int C,
                                 void* stream);
/* pool-code bytes (round 5): the forward pooling also writes ONE byte per pooled element ([N][ceil(H/2)][ceil(W/2)][C]: bits 1:0 = window
 * position of the first maximum in scan order (0,0) (0,1) (1,0) (1,1), bits 5:2 = "input at position q > 0"), and the backward reads that byte
 * instead of the pool's four inputs (vgg_osvos.py:140 ceil-mode max-pool + the ReLU backward of the producing convolution + the side-branch
 * add, as osvos_maxpool2x2_bwd_bf16act): same results bit for bit, 1 byte instead of 8 read per pooled element. */
int osvos_maxpool2x2_bf16act_code(const void* x_bf16, void* y_bf16, void* code, int N, int H, int W, int C, void* stream);
int osvos_maxpool2x2_bwd_bf16act_code(const void* code, const void* dy_bf16, const void* dside_bf16, void* dx_bf16, int N, int H, int W, int C,
                                      void* stream);
/* The trunk convolution of the bf16-store mode WITH its fused epilogues, as osvos_net_forward / osvos_net_backward launch it (round 6: op-level
 * access for the parity tests of every tile, vgg_osvos.py:136-145 conv -> ReLU [-> MaxPool2d(2, stride=2, ceil_mode=True)] and its autograd):
 *   x_bf16, y_bf16: bf16 NHWC, dense (channel stride = channel count); wpk from osvos_pack_fwd / osvos_pack_dgrad with OSVOS_F32_BF16MFMA
 *   mask_bits (optional): ReLU mask of a data gradient as ONE bit per element, [N][H][W][Cout/32] 32-bit words, bit b of word g = channel 32 g + b
 *   y_bits (optional): the same for the result (what a later data gradient is masked with)
 *   pooled_bf16 + pool_code (optional; need relu, no mask): the 2x2 ceil-mode max-pool of the result and its code bytes (see above)
 *   tile: osvos_conv3x3_bf16io_tiles() ids (+100: XCD-local block order) or -1 = automatic; 36 / 37 need Cin == 64. */
int osvos_conv3x3_bf16act_fused(const void* x_bf16, const void* wpk, const float* bias, const void* mask_bits, void* y_bf16, void* y_bits,
                                void* pooled_bf16, void* pool_code, int N, int H, int W, int Cin, int Cout, int relu, int tile, void* stream);
int osvos_conv3x3_bf16io_tiles(int* tiles, int max);
/* pieces per operand of the OSVOS_F32_X3 kernels called from THIS host thread through the op-level entry points (osvos_conv3x3,
 * osvos_conv3x3_x3*, osvos_conv3x3_wgrad, osvos_pack_conv3x3_x3 with that dtype): 3 (default: three bf16 pieces, six products, fp32-grade),
 * 2 (two bf16 pieces, three products; what OSVOS_FLAG_X3_TWO_PIECES selects for the osvos_net_* calls, which set and restore it themselves)
 * or 22 (two FP16 pieces under block exponents, three products; OSVOS_FLAG_X3_HALF_PIECES -- a pack made under 22 is in the FP16-pair
 * format and must be read under 22). */
int osvos_set_x3_pieces(int pieces);
int osvos_nchw_to_nhwc_bf16copy(const float* src, void* dst, void* dst_bf16, int N, int C, int H, int W, int cpad, void* stream);
int osvos_maxpool2x2_bf16copy(const float* x, float* y, void* y_bf16, int N, int H, int W, int C, void* stream);
int osvos_maxpool2x2_bwd_bf16copy(const float* x, const float* dy, const float* dside, float* dx, void* dx_bf16,
                                  int N, int H, int W, int C, void* stream);

/* ---- 3x3 weight gradient (weight/bias half of aten::convolution_backward) -----------------
 * dW[co,ci,r,s] = sum_{n,h,w} dY[n,h,w,co] * x[n,h+r-1,w+s-1,ci];  db[co] = sum dY
 *   x: NHWC stride Cin_s (only ci < Cin used); dy: NHWC stride Cout_s (already ReLU-masked)
 *   ws: workspace of osvos_wgrad_ws_bytes(); dw: fp32 OIHW [Cout,Cin,3,3]; db: fp32 [Cout] or NULL
 *   accumulate != 0: dw/db += result (gradient accumulation) else overwritten */
size_t osvos_wgrad_ws_bytes(int N, int H, int W, int Cin, int Cout, int dtype);
int osvos_conv3x3_wgrad(const void* x, const void* dy, void* ws, float* dw, float* db,
                        int N, int H, int W, int Cin, int Cin_s, int Cout, int Cout_s,
                        int accumulate, int dtype, void* stream);

/* ---- 2x2/2 max-pool, ceil_mode (aten::max_pool2d_with_indices, vgg_osvos.py:140) ---------- */
int osvos_maxpool2x2(const void* x, void* y, int N, int H, int W, int C, int dtype, void* stream);
/* fused backward: dx = relu_mask(x) * ( route(dy, first max of the window in scan order) + dside )
 * x is the (post-ReLU) pool input, dside (NULL ok) the gradient from the side branch. */
int osvos_maxpool2x2_bwd(const void* x, const void* dy, const void* dside, void* dx,
                         int N, int H, int W, int C, int dtype, void* stream);

/* ---- side-output / fuse head (vgg_osvos.py:68-72: score_dsn 1x1, upscale_/upscale transposed
 *      convs, center_crop (osvos_layers.py:51-56), torch.cat, fuse 1x1) ----------------------
 * Commuted form: valid when upscale[i].weight is diagonal with one shared
 * k x k filter; the caller checks that with osvos_deconv_diag_check and must refuse otherwise.
 *   prep:   NHWC [N,h,w,16] output of side_prep[i]
 *   wd/bd:  score_dsn[i] weight[16] / bias[1];  wf: fuse.weight[16*i .. 16*i+15]
 *   score, fpart: fp32 [N,h,w] low-resolution maps */
int osvos_head_lowres(const void* prep, const float* wd, const float* bd, const float* wf,
                      float* score, float* fpart, int N, int h, int w, int dtype, void* stream);
/* outs[0..3] = crop(upscale_[i](score_i)), outs[4] = fuse.bias + sum_i crop(up_i(fpart_i)); all
 * fp32 NCHW [N,1,H,W].  score/fpart/f1/f16: host arrays of 4 device pointers (filters k x k,
 * k = 4,8,16,32; f1 = upscale_[i].weight[0,0], f16 = upscale[i].weight[0,0]); hs/ws: host int[4]. */
int osvos_head_upsample(const float* const* score, const float* const* fpart,
                        const float* const* f1, const float* const* f16, const float* fuse_bias,
                        float* const* outs, int N, int H, int W, const int* hs, const int* ws, void* stream);
/* backward of both, per scale: dprep[N,h,w,16] (NHWC, dtype) = wf*up^T(dfused) + wd*up_^T(dside);
 * writes per-workgroup partial sums of the weight/bias gradients to acc (double[OSVOS_HEAD_MAX_BLOCKS][34] at most:
 * {dwf[16], dwd[16], dbd, spare} per launched workgroup; the whole-network call sums them).
 * dside / dfused: fp32 NCHW [N,1,H,W] or NULL (treated as zero). */
#define OSVOS_HEAD_MAX_BLOCKS 1024
int osvos_head_bwd(const void* prep, const float* dside, const float* dfused,
                   const float* f1, const float* f16, const float* wd, const float* wf,
                   void* dprep, double* acc, int N, int H, int W, int h, int w, int scale_idx,
                   int dtype, void* stream);
/* max |off-diagonal| and max |w[c,c]-w[0,0]| of a [16,16,k,k] deconv weight -> out[2] (device) */
int osvos_deconv_diag_check(const float* w, int C, int k, float* out2, void* stream);

/* ---- class-balanced BCE with logits (osvos_layers.py:19-48) ------------------------------
 * out/label fp32, `count` elements over N images.  mode 0: size_average, 1: batch_average,
 * 2: neither.  loss: fp32[1]; grad: fp32[count] = dLoss/dOut (upstream 1) or NULL;
 * scratch: 32 bytes, zeroed by this call.  Tensors that are not 16-byte aligned (offset views) are swept element-wise: slower, same numbers. */
int osvos_cbce(const float* out, const float* label, float* loss, float* grad, void* scratch,
               long count, int N, int mode, void* stream);
/* The same loss as one step of the training loops sees it (train_online.py:127-141, train_parent.py:143-163): `loss` is the plain
 * loss (what the reference adds to running_loss), `running` (device fp32[1] or NULL) += loss inside the final kernel, and
 * grad = grad_scale * dLoss/dOut with grad_scale = the upstream gradient the reference's `loss /= nAveGrad; loss.backward()` hands the
 * loss (1/nAveGrad, times (1 - epoch/nEpochs) for the parent loop's side heads) -- rounded like osvos_cbce followed by osvos_scale,
 * but without the separate scale / add / divide launches between the loss and the head's backward. */
int osvos_cbce_step(const float* out, const float* label, float* loss, float* grad, void* scratch,
                    long count, int N, int mode, float grad_scale, float* running, void* stream);
/* Several heads against ONE label (train_parent.py:143-147: the five losses of a micro-batch) in three launches instead of 3 per head: the class
 * counts of the label are formed once.  outs / losses / grads / running: arrays of n_heads (1..8) device pointers (grads, running and their entries
 * may be NULL); grad_scales: n_heads HOST floats; scratch: 32 x n_heads bytes, zeroed by this call.  Per head identical to osvos_cbce_step. */
int osvos_cbce_step_multi(const float* const* outs, const float* label, float* const* losses, float* const* grads, void* scratch,
                          long count, int N, int mode, int n_heads, const float* grad_scales, float* const* running, void* stream);
/* General form (the three calls above forward to it with flags = 0, counts = NULL).
 *   flags & OSVOS_CBCE_PER_IMAGE: each of the N images is its own reference batch of ONE -- class weights from its own label, size_average
 *     divides by the elements of one image, batch_average by 1 -- and `loss` receives the SUM of the N losses: the nAveGrad micro-batches of
 *     an accumulation window (train_online.py:116-149: cbce(..., batch of 1); loss /= nAveGrad; backward(); running_loss += loss) in ONE call
 *     on a batch-N forward, with grad_scale = 1 / nAveGrad.
 *   counts != NULL: device fp32[3] = {n_pos, n_total, n_images} of the GLOBAL batch these tensors are a shard of (the count exchange of a
 *     batch sharded over ranks; layers/osvos_layers.py:28-34,43-46 count over the whole input tensor): weights and divisors come
 *     from them, no count sweep runs.  Summed over the shards the losses / gradients are those of the whole batch.  Not with PER_IMAGE.
 *   scratch: osvos_cbce_scratch_bytes(n_heads, N, flags) bytes, zeroed by the call -- or, with flags & OSVOS_CBCE_SCRATCH_ZEROED, already zero by
 *     the caller's promise (no memset is enqueued).  Every call LEAVES the scratch zero (its last workgroup forms the losses and clears what it
 *     read), so a buffer that was zeroed once can be handed to call after call on one stream. */
#define OSVOS_CBCE_PER_IMAGE 1
#define OSVOS_CBCE_SCRATCH_ZEROED 2
size_t osvos_cbce_scratch_bytes(int n_heads, int N, int flags);
int osvos_cbce_step_ex(const float* const* outs, const float* label, float* const* losses, float* const* grads, void* scratch, long count,
                       int N, int mode, int flags, const float* counts, int n_heads, const float* grad_scales, float* const* running,
                       void* stream);
/* y[i] = x[i] * (*scalar)   (loss.backward() chain rule with a device-resident upstream grad) */
int osvos_scale(const float* x, const float* scalar, float* y, long count, void* stream);

/* ---- whole network ------------------------------------------------------------------------
 * One call = OSVOS.forward (vgg_osvos.py:59-74) / its autograd backward.
 *   params: host array of 52 device pointers, state_dict order, fp32, contiguous
 *   wbuf:   persistent buffer of osvos_net_wbuf_bytes(): packed weights (refresh with
 *           osvos_net_pack whenever a parameter changed, e.g. after optimizer.step())
 *   ws:     per-call workspace of osvos_net_ws_bytes(); forward leaves the activations the
 *           backward needs in it (and, in the bf16-store and f32x3 modes, the SIGN BITS of the activations that mask a later data
 *           gradient -- one 32-bit word per pixel and 32 channels, csrc/maskbits.h -- so the same ws must go to the backward untouched)
 *   outs:   host array of 5 device pointers, fp32 [N,1,H,W] */
size_t osvos_net_wbuf_bytes(int dtype);
size_t osvos_net_ws_bytes(int N, int H, int W, int dtype);
/* workspace of a forward that will never be followed by osvos_net_backward (inference): a prefix of ws */
size_t osvos_net_ws_bytes_infer(int N, int H, int W, int dtype);
int osvos_net_pack(const float* const* params, void* wbuf, int dtype, int with_dgrad, void* stream);
int osvos_net_forward(const float* x_nchw, const void* wbuf, void* ws, float* const* outs,
                      int N, int H, int W, int dtype, void* stream, void* aux_stream /* NULL ok: see backward */);
/* douts: host array of 5 device pointers (NULL = no gradient for that head).
 * grads: host array of 52 device pointers (NULL entries are skipped; deconv weights are frozen in
 *        both reference scripts -- train_online.py:84-85 -- and are never written).
 * dx_nchw: fp32 [N,3,H,W] or NULL.  accumulate != 0: grads += instead of overwrite.
 * aux_stream: NULL, or a second stream owned by the caller: the weight-gradient kernels are then
 *   enqueued on it, concurrently with the data-gradient kernel of the same layer (fork/join with
 *   events; everything has joined `stream` again when the call returns).
 * aux2_stream: NULL, or a third stream for the bandwidth-bound slab reduces of the weight gradients, so they
 *   do not sit between two MFMA weight-gradient kernels on aux_stream. */
int osvos_net_backward(const void* wbuf, void* ws, const float* const* douts, float* const* grads,
                       float* dx_nchw, int N, int H, int W, int dtype, int accumulate, void* stream, void* aux_stream,
                       void* aux2_stream);
/* makes `stream` wait for everything enqueued so far on aux_stream / aux2_stream (NULL = skip): the join a backward with
 * OSVOS_FLAG_DEFER_JOIN left out */
int osvos_net_join(void* stream, void* aux_stream, void* aux2_stream);
/* Gradient-ready events for overlapping the data-parallel all-reduce with the rest of the backward (extends
 * train_parent.py:163-172; SURVEY 8e "overlap with the last micro-batch's backward, bucket order = reverse layer order").
 * Arms the NEXT osvos_net_backward call made by THIS host thread: events[k] (hipEvent_t handles owned by the caller, k < 7) is
 * recorded on whichever stream writes the last gradient of group k, right after that write:
 *   0 = score_dsn + fuse, 1 = side_prep (all four), 2 = stages.4, 3 = stages.3, 4 = stages.2, 5 = stages.1, 6 = stages.0
 * -- the order in which a backward completes them.  A communication stream that waits on events[k] may reduce group k's gradients
 * while the shallower layers are still being computed.  n = 0 disarms.  Entries may be NULL. */
#define OSVOS_NGRAD_GROUPS 7
int osvos_net_arm_grad_events(void* const* events, int n);
/* ---- RCCL through the C ABI: the gradient exchange of the data-parallel loop (one sum of the flat gradient buffer per
 *      optimizer step; extends train_parent.py:163-172) without torch.distributed.  librccl is bound with dlopen at the first call.
 *   osvos_comm_unique_id: on ONE rank; ship the 128 bytes to the others by any means (file, socket, torch.distributed broadcast).
 *   osvos_comm_init: collective over all `world` ranks, on the calling process's current HIP device; *comm is the only library-owned
 *     handle of this ABI (opaque; osvos_comm_destroy releases it).
 *   osvos_comm_allreduce_f32: in-place sum of buf[0..count) over the ranks, enqueued on `stream` (no host synchronisation).
 *   osvos_comm_allreduce_chunks_f32: the same as n chunks buf[first[k] .. first[k] + count[k]), chunk k enqueued on comm_stream behind
 *     ready_events[k] (hipEvent_t, NULL = no wait): with the events of osvos_net_arm_grad_events and the chunks in the backward's
 *     completion order, the deep layers' gradients travel while the shallow ones are still being computed.
 *   Return: 0, < 0 argument / loader error, hipError_t, or 1000 + ncclResult_t. */
#define OSVOS_COMM_ID_BYTES 128
int osvos_comm_unique_id(void* id128);
int osvos_comm_init(void** comm, int rank, int world, const void* id128);
int osvos_comm_allreduce_f32(void* comm, float* buf, size_t count, void* stream);
int osvos_comm_allreduce_f64(void* comm, double* buf, size_t count, void* stream);   /* sum of doubles: class counts, loss statistics (exact) */
int osvos_comm_allreduce_chunks_f32(void* comm, float* buf, const size_t* first, const size_t* count, void* const* ready_events, int n,
                                    void* comm_stream);
int osvos_comm_destroy(void* comm);
/* byte offset / element count of a saved activation inside ws (tests): which = 0..12 trunk conv
 * outputs, 13..16 pooled inputs of stages 1-4, 17..20 side_prep outputs, 21 NHWC input.  (fp32 elements; with dtype
 * OSVOS_F32_BF16MFMA the trunk tensors 0..16 are bf16 unless OSVOS_BF16_STORE=0.) */
int osvos_net_ws_query(int N, int H, int W, int dtype, int which, size_t* offset, size_t* elems, int* channels, int* h, int* w);
/* storage format of trunk tensor `which` (0..16) for `dtype`: 0 fp32 NHWC, 1 bf16 NHWC (OSVOS_F32_BF16MFMA store mode) */
int osvos_net_ws_format(int dtype, int which);

/* ---- training-time input pipeline (dataloaders/davis_2016.py:99-106 + custom_transforms.py: flip, ScaleNRotate, ToTensor) ------
 * One frame: uint8 BGR [H][W][3] (+ uint8 label [H][W] or NULL) -> float32 image [3][H][W] = (flipped, warped) frame minus mean3,
 * float32 gt [1][H][W] = label / max(label.max(), 1e-8) (nearest warp for 0/1 masks, bicubic for soft ones, decided on the device).
 * Minv: HOST pointer to the dst->src affine cv::warpAffine derives from getRotationMatrix2D (6 doubles), NULL = no warp.
 * scratch: 2 unsigned on the device.  OpenCV's fixed-point coordinate / 5-bit cubic table algorithm, border value 0. */
int osvos_augment_frame(const unsigned char* img_bgr, const unsigned char* label, const float* mean3, int flip, const double* Minv,
                        float* out_img, float* out_gt, void* scratch, int H, int W, void* stream);

/* ---- result writer / evaluator (train_online.py:181-189) -----------------------------------
 * osvos_mask_to_bytes: per image p = sigmoid(logit), then scipy<=1.1 imsave's min-max byte scaling
 *   uint8((p - min) * 255/(max - min) clipped + 0.5); scratch = 2 N unsigned.  The host side writes the PNG.
 * osvos_mask_iou_counts: per image {|P & G|, |P | G|} with P = logit > logit_threshold, G = gt > 0.5 (DAVIS region measure J);
 *   counts = 2 N unsigned long long. */
int osvos_mask_to_bytes(const float* logits, unsigned char* out, void* scratch, long count, int N, void* stream);
int osvos_mask_iou_counts(const float* logits, const float* gt, void* counts, long count, int N, float logit_threshold, void* stream);

/* ---- fused SGD (torch.optim.SGD semantics, train_online.py:79-88,147) ----------------------
 * for each i: d = g + wd*p; buf = first ? d : momentum*buf + d; p -= lr*buf      (flat tensors) */
int osvos_sgd_step(float* p, const float* g, float* buf, long count, float lr, float momentum,
                   float weight_decay, int first, void* stream);
/* the same update for a whole list of tensors in one launch (per-tensor lr / weight decay = the reference's 8 SGD
 * parameter groups); host arrays of n device pointers / counts / rates; tensors whose group has lr = 0 and no
 * weight decay can simply be left out.  `first` != 0: the momentum buffers are initialised (buf = d). */
#define OSVOS_SGD_MAX_TENSORS 64
int osvos_sgd_step_multi(float* const* params, const float* const* grads, float* const* bufs, const long* counts,
                         const float* lrs, const float* wds, int n, float momentum, int first, void* stream);

/* ---- opt-in launch profiler (bench.py only): hipEvent pairs around every kernel family that
 * osvos_net_forward/backward launches, recorded on the caller's stream.  Families: 0 conv3x3
 * forward launches, 1 backward conv regions (data-gradient kernel || weight-gradient kernel +
 * slab reduce of one layer, fork to join), 2-3 spare.
 * osvos_prof_stop fills ms[4] (sum of launch durations), flops[4] (algorithmic FLOPs of those
 * launches: 2*N*H*W*Cout*9*Cin) and count[4]; synchronise the stream before calling it. */
int osvos_prof_start(int max_records);
int osvos_prof_stop(double* ms, double* flops, long* count);
/* between start and stop: paused != 0 suspends recording (no events are enqueued), 0 resumes it; returns the previous state */
int osvos_prof_pause(int paused);

/* ---- debug references (plain one-thread-per-output kernels, used only by tests) ----------- */
int osvos_debug_conv3x3_naive(const float* x, const float* w_oihw, const float* bias, float* y,
                              int N, int H, int W, int Cin, int Cin_s, int Cout, int relu, void* stream);
int osvos_debug_mfma_layout(float* out /* 4*64*16 floats */, void* stream);
int osvos_debug_mfma_peak(float* out /* blocks*256 floats */, int blocks, int iters, void* stream);
/* register-only loop of v_mfma_f32_32x32x16_bf16 (blocks x 8 waves x iters x 8 MFMAs) on the caller's operand bits (seed: 2048 bytes of bf16):
 * the rate the bf16 matrix pipe SUSTAINS on this chip -- 2.48 PFLOP/s at 2.37 GHz on zeros, 1.83 at 1.81 GHz on noise (bench.py reports it live) */
int osvos_debug_mfma_peak_bf16(const void* seed, float* out /* blocks*512 floats */, int blocks, int iters, void* stream);
/* bf16-store mode: conv1_1's weight gradient on the bf16 matrix pipe (1, default) or on the exact fp32 skinny kernel fed with the bf16 dY
 * (0); returns the previous setting.  Tests compare the two. */
int osvos_debug_set_c3_bf16(int on);
/* LDS-DMA layout probe (buffer_load_dwordx4 ... lds): out[8*64*4] = LDS image after 4 waves x 2 DMA instructions; the test
 * pins "lane l of instruction j lands in 16-byte slot 64 j + l, out-of-range lanes land as zeros" (tests/test_gpu_ops.py) */
int osvos_debug_lds_dma(const float* src, int ngroups, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
