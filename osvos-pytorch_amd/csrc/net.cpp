// Whole-network orchestration: OSVOS.forward (reference vgg_osvos.py:59-74) and its backward as
// two C calls that only enqueue kernels on the caller's stream (no allocation, no sync, so both
// are hipGraph-capturable).  The caller owns `wbuf` (packed parameters) and `ws` (activations +
// gradient scratch); their layouts are defined here and nowhere else.
#include <stdlib.h>
#include <string.h>

#include "kernels.h"
#include "prof.h"

namespace {
inline int streamk_mode();      // (defined with the other process-cached switches below)

constexpr int kStageN[5] = {2, 2, 3, 3, 3};
constexpr int kStageC[5] = {64, 128, 256, 512, 512};
constexpr int kNumTrunk = 13;
constexpr int kNumConv = 17;          // 13 trunk + 4 side_prep
constexpr int kInPad = 8;             // conv1_1 input channels padded 3 -> 8 (one K chunk)

struct ConvDesc {
  int stage, cin, cin_s, cout, w_param, b_param;
};

void conv_table(ConvDesc* d) {
  int l = 0, cin = 3;
  for (int si = 0; si < 5; ++si)
    for (int j = 0; j < kStageN[si]; ++j) {
      d[l] = ConvDesc{si, cin, cin == 3 ? kInPad : cin, kStageC[si], 8 + 2 * l, 9 + 2 * l};
      cin = kStageC[si];
      ++l;
    }
  for (int i = 0; i < 4; ++i) d[kNumTrunk + i] = ConvDesc{i + 1, kStageC[i + 1], kStageC[i + 1], 16, 34 + 2 * i, 35 + 2 * i};
}

int last_of_stage(int si) {
  int l = -1;
  for (int s = 0; s <= si; ++s) l += kStageN[s];
  return l;
}

struct WbufLayout {
  size_t fwd[kNumConv], dgrad[kNumConv], bias[kNumConv];
  size_t fwd3[kNumConv], dgrad3[kNumConv];      // OSVOS_F32_X3: pre-split bf16x3 packs of the layers the f32x3 kernels take ((size_t)-1: none)
  size_t wd[4], bd[4], wf, bf, f1[4], f16[4];
  size_t weff[4];            // generic head only: Weff_i[16][k*k] (head_generic.hip)
  size_t wup[4];             // generic head only: copy of upscale[i].weight [16][16][k][k] (the backward forms fuse / upscale gradients from it)
  size_t total;
};

WbufLayout wbuf_layout(int dtype) {
  WbufLayout L;
  ConvDesc d[kNumConv];
  conv_table(d);
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  for (int l = 0; l < kNumConv; ++l) {
    L.fwd[l] = take(osvos_wpack_bytes(d[l].cout, d[l].cin_s, dtype));
    L.dgrad[l] = take(osvos_wpack_dgrad_bytes(d[l].cout, d[l].cin, dtype));
    L.bias[l] = take(sizeof(float) * d[l].cout);
    L.fwd3[l] = L.dgrad3[l] = (size_t)-1;
    if (dtype == OSVOS_F32_X3) {
      if (d[l].cin_s % 16 == 0 && d[l].cin == d[l].cin_s) L.fwd3[l] = take(osvos_wpack_x3_bytes(d[l].cout, d[l].cin));
      if (d[l].cout % 16 == 0) L.dgrad3[l] = take(osvos_wpack_x3_bytes(d[l].cin, d[l].cout));
    }
  }
  for (int i = 0; i < 4; ++i) { L.wd[i] = take(16 * sizeof(float)); L.bd[i] = take(sizeof(float)); }
  L.wf = take(64 * sizeof(float));
  L.bf = take(sizeof(float));
  for (int i = 0; i < 4; ++i) { const int k = 4 << i; L.f1[i] = take(sizeof(float) * k * k); L.f16[i] = take(sizeof(float) * k * k); }
  for (int i = 0; i < 4; ++i) { const int k = 4 << i; L.weff[i] = take(sizeof(float) * 16 * k * k); }
  for (int i = 0; i < 4; ++i) { const int k = 4 << i; L.wup[i] = take(sizeof(float) * 256 * k * k); }
  L.total = off;
  return L;
}

// bf16-ONLY trunk tensors: in the bf16-MFMA mode activations, pooled tensors and their gradients are stored as bf16 and nothing
// fp32 is written for them (default; OSVOS_BF16_STORE=0 keeps fp32 tensors and rounds while staging).  Producers write half the
// bytes, consumers read half the bytes; pooling and its backward run on bf16; bias / skinny weight gradients are formed from the
// bf16 tensors.  Measured: 666 -> 746 frames/s (batch 12), 374 -> 415 (batch 1).
inline bool use_store(int dtype) {
  static const bool on = [] { const char* e = getenv("OSVOS_BF16_STORE"); return !(e && e[0] == '0'); }();
  return dtype == OSVOS_F32_BF16MFMA && on;
}

struct WsLayout {
  int hs[5], ws[5];
  size_t xin, act[kNumTrunk], pooled[5], prep[4], score[4], fpart[4];
  size_t conv_part;          // split-K partial sums of the small deep layers (forward prefix: inference uses it too)
  size_t sk_ws;              // f32x3: stream-K workspace of the main-stream convolutions (tickets + partial slots; (size_t)-1 = none)
  size_t side_part[4];       // the same for the side_prep convolutions, which run on the aux stream beside the trunk (own buffers)
  size_t dy[kNumTrunk], dpool[5], dside[4], dprep[4], wgrad[kNumConv], acc, dxin;
  size_t gbuf[4];            // generic head only: tap-indexed reductions G_i[16][k*k] + G1_i[k*k], doubles
  // the bf16 trunk tensors of the bf16-store mode (dtype OSVOS_F32_BF16MFMA, OSVOS_BF16_STORE=1): the SAME offsets as the fp32 names above
  // (nothing fp32 is written for them); 0 / unused otherwise.  (Rounds 1-3 could also keep bf16 COPIES next to fp32 tensors --
  // OSVOS_BF16_SHADOW, measured a net loss at batch 12 -- removed in round 4.)
  size_t xin_b, act_b[kNumTrunk], pooled_b[5], dy_b[kNumTrunk], dpool_b[5], dside_b[4], dprep_b[4];
  // sign bits of the activations that later serve as ReLU masks (maskbits.h; (size_t)-1 = none): [N][h][w][cout / 32] words
  size_t bits[kNumTrunk];
  // bf16-store mode: one code byte per pooled element, written by the forward's pooling (fused epilogue or kernel), read by maxpool_bwd
  // instead of the pool's input (pool.hip; (size_t)-1 = none)
  size_t pool_code[5];
  size_t fwd_total, total;
};

// One-bit ReLU masks (maskbits.h): the forward writes the sign bits of every activation a data gradient is later masked with, the data
// gradient reads one word per (pixel, 32 channels) instead of the activation itself.  bf16-store mode (OSVOS_MASK_BITS=0: the activation).
inline bool use_mask_bits(int dtype) {
  static const bool on = [] { const char* e = getenv("OSVOS_MASK_BITS"); return !(e && e[0] == '0'); }();
  return on && (use_store(dtype) || dtype == OSVOS_F32_X3);
}
// act[l] masks the data gradient of layer l + 1 when both are in the same stage; stage 4's last activation masks its side branch's.
// f32x3 (fp32 tensors): only where the f32x3 kernel produces the activation in ONE launch -- not conv1_1
// (exact kernel) and not stage 4, whose forward launches are cut along K (bits would pin them to one K range)
inline bool act_is_a_mask(const ConvDesc* d, int l, int dtype) {
  const bool is = (l + 1 < kNumTrunk && d[l + 1].stage == d[l].stage) || l == kNumTrunk - 1;
  if (dtype == OSVOS_F32_X3) return is && l >= 1 && d[l].stage <= 3;
  return is;
}

// A/B switch of the round-5 pool-code bytes (OSVOS_POOL_CODE=0: maxpool_bwd recomputes the argmax from the pool's input, as in rounds 1-4)
inline bool use_pool_code() {
  static const bool on = [] { const char* e = getenv("OSVOS_POOL_CODE"); return !(e && e[0] == '0'); }();
  return on;
}

WsLayout ws_layout(int N, int H, int W, int dtype) {
  WsLayout L;
  memset(&L, 0, sizeof(L));
  for (int l = 0; l < kNumTrunk; ++l) L.bits[l] = (size_t)-1;
  for (int si = 0; si < 5; ++si) L.pool_code[si] = (size_t)-1;
  const size_t es = osvos_elem(dtype);
  L.hs[0] = H; L.ws[0] = W;
  for (int i = 1; i < 5; ++i) { L.hs[i] = (L.hs[i - 1] + 1) / 2; L.ws[i] = (L.ws[i - 1] + 1) / 2; }
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  ConvDesc d[kNumConv];
  conv_table(d);
  const bool store = use_store(dtype);
  const size_t te = store ? 2 : es;          // element size of the trunk tensors
  L.xin = take(es * N * H * W * kInPad);
  if (store) L.xin_b = take((size_t)2 * N * H * W * kInPad);
  for (int l = 0; l < kNumTrunk; ++l) {
    const int si = d[l].stage;
    const size_t e = (size_t)N * L.hs[si] * L.ws[si] * d[l].cout;
    L.act[l] = take(te * e);
    L.act_b[l] = store ? L.act[l] : 0;
    L.bits[l] = (use_mask_bits(dtype) && act_is_a_mask(d, l, dtype)) ? take(e / 8) : (size_t)-1;
  }
  for (int si = 1; si < 5; ++si) {
    const size_t e = (size_t)N * L.hs[si] * L.ws[si] * kStageC[si - 1];
    L.pooled[si] = take(te * e);
    L.pooled_b[si] = store ? L.pooled[si] : 0;
    L.pool_code[si] = (store && use_pool_code()) ? take(e) : (size_t)-1;
  }
  for (int i = 0; i < 4; ++i) {
    const size_t npix = (size_t)N * L.hs[i + 1] * L.ws[i + 1];
    L.prep[i] = take(es * npix * 16);
    L.score[i] = take(sizeof(float) * npix);
    L.fpart[i] = take(sizeof(float) * npix);
  }
  {
    size_t mx = 0;
    for (int l = 0; l < kNumTrunk; ++l)
      if (d[l].cin >= 256) {
        const size_t b = osvos_conv3x3_splitk_ws_bytes_f32(N, L.hs[d[l].stage], L.ws[d[l].stage], d[l].cout > d[l].cin ? d[l].cout : d[l].cin);
        if (b > mx) mx = b;
      }
    L.conv_part = take(mx);
    // stream-K is opt-in (OSVOS_X3_STREAMK, read once per process): its 64 MiB of tickets + partial tiles exist only in layouts that use them
    L.sk_ws = (dtype == OSVOS_F32_X3 && streamk_mode() >= 1) ? take(osvos_conv3x3_f32x3_streamk_ws_bytes()) : (size_t)-1;
    // side_prep[i]: Cout = 16 on a small frame is a handful of workgroups walking K = 9 Cin serially (88 us for 0.24 GFLOP at
    // 30 x 54) -- and the last one sits exposed between conv5_3 and the head.  K splits turn it into a full-chip launch.
    for (int i = 0; i < 4; ++i)
      L.side_part[i] = kStageC[i + 1] >= 256 ? take(osvos_conv3x3_splitk_ws_bytes_f32(N, L.hs[i + 1], L.ws[i + 1], 16)) : (size_t)-1;
  }
  L.fwd_total = off;          // everything above is all an inference-only forward touches
  for (int i = 0; i < 4; ++i) {
    const size_t npix = (size_t)N * L.hs[i + 1] * L.ws[i + 1];
    L.dprep[i] = take(es * npix * 16);
    L.dprep_b[i] = store ? take((size_t)2 * npix * 16) : 0;
    L.dside[i] = take(te * npix * kStageC[i + 1]);
    L.dside_b[i] = store ? L.dside[i] : 0;
  }
  // one gradient buffer per trunk conv output (dLoss/d act[l], ReLU mask applied) and per pooled
  // tensor: no buffer is ever rewritten inside one backward, so the weight-gradient stream can trail
  // the data-gradient stream by any number of layers without write-after-read hazards
  for (int l = 0; l < kNumTrunk; ++l) {
    const size_t e = (size_t)N * L.hs[d[l].stage] * L.ws[d[l].stage] * d[l].cout;
    L.dy[l] = take(te * e);
    L.dy_b[l] = store ? L.dy[l] : 0;
  }
  for (int si = 1; si < 5; ++si) {
    L.dpool[si] = take(te * N * L.hs[si] * L.ws[si] * kStageC[si - 1]);
    L.dpool_b[si] = store ? L.dpool[si] : 0;
  }
  // one slab workspace per layer: the slab reduce of layer l runs on its own stream while the partial
  // kernel of the next layer already refills another buffer
  for (int l = 0; l < kNumConv; ++l) {
    const int si = d[l].stage;
    L.wgrad[l] = take(osvos_wgrad_ws_bytes(N, L.hs[si], L.ws[si], d[l].cin_s, d[l].cout, dtype));
  }
  L.acc = take(sizeof(double) * (5 * OSVOS_HEAD_MAX_BLOCKS * 34));   // head_bwd partials: 4 scales + fuse bias
  L.dxin = take(es * N * H * W * 4);
  for (int i = 0; i < 4; ++i) { const int k = 4 << i; L.gbuf[i] = take(sizeof(double) * 17 * k * k); }
  L.total = off;
  return L;
}

// events for the fork/join of the two-stream backward: created once per host thread, reused round-robin
struct EventPool {
  hipEvent_t ev[128];      // (round-robin: an event recorded early in a backward and waited for late must not come around in between -- ~45 per call)
  int n = 0, cur = 0;
  hipEvent_t next() {
    if (n < 128) {
      hipEvent_t e;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
        osvos_set_error("net_backward: hipEventCreate failed");
        return nullptr;
      }
      ev[n++] = e;
      return e;
    }
    hipEvent_t e = ev[cur];
    cur = (cur + 1) % 128;
    return e;
  }
};
EventPool& event_pool() {
  static thread_local EventPool p;
  return p;
}

// f32x3: weights pre-split once per pack (default) or re-split by every workgroup from the fp32 pack (OSVOS_X3_PRESPLIT=0; bit-identical)
inline bool use_presplit() {
  static const bool on = [] { const char* e = getenv("OSVOS_X3_PRESPLIT"); return !(e && e[0] == '0'); }();
  return on;
}

// 3x3 conv on the main stream: fp32 launches may be cut along K (split-K, partial sums in `part`) when the layer
// is too small to balance across 256 CUs; the bf16-MFMA dtype goes through the public entry point
// (x_b: bf16 copy of x, preferred when present; y_b: where the bf16 copy of y goes, NULL = none)
// (mask_b: bf16 mask, takes precedence over the fp32 `mask`; y may be NULL in the bf16 modes when y_b is given)
// (epi: fused pooling epilogues, f32x3 only -- fuse_pool() says when the caller may ask for them)
inline int conv_main(const void* x, const void* x_b, const void* wpk, const float* bias, const void* mask, const void* mask_b, void* y, void* y_b,
                     int N, int h, int w, int cin, int cout, int y_cs, int relu, int dtype, void* part, hipStream_t stream, const void* wpk3 = nullptr,
                     const ConvEpi* epi = nullptr, const void* mask_bits = nullptr, void* y_bits = nullptr, void* pooled_b = nullptr, void* sk_ws = nullptr,
                     void* pool_code = nullptr) {
  if (dtype == OSVOS_F32_X3 && osvos_conv3x3_f32x3_applicable(cin, cout, y_cs)) {    // three-way bf16 split on the bf16 matrix pipe
    ConvEpi e2;
    if (epi != nullptr) e2 = *epi;
    e2.mask_bits = reinterpret_cast<const unsigned*>(mask_bits);
    e2.y_bits = reinterpret_cast<unsigned*>(y_bits);
    e2.sk_ws = sk_ws;
    const bool any = epi != nullptr || mask_bits != nullptr || y_bits != nullptr || sk_ws != nullptr;
    // (with a pre-split pack the fp32 pack of the layer is not even built -- osvos_net_pack -- so it is not handed over either)
    return osvos_conv3x3_f32x3_epi((const float*)x, (use_presplit() && wpk3) ? nullptr : (const float*)wpk, use_presplit() ? wpk3 : nullptr, bias,
                                   (const float*)mask, (float*)y, N, h, w, cin, cout, y_cs, relu, -1, 0, part, any ? &e2 : nullptr, stream);
  }
  if (dtype == OSVOS_F32 || dtype == OSVOS_F32_X3)
    return osvos_conv3x3_f32_ws((const float*)x, (const float*)wpk, bias, (const float*)mask, (float*)y, N, h, w, cin, cout, y_cs,
                                relu, -1, part, stream);
  return osvos_conv3x3_bf16mfma_bits(x_b ? x_b : x, x_b ? 1 : 0, wpk, bias, mask_b ? mask_b : mask, mask_b ? 1 : 0, (const unsigned*)mask_bits, (float*)y, y_b,
                                     (unsigned*)y_bits, pooled_b, N, h, w, cin, cout, y_cs, relu, -1, stream, pooled_b ? pool_code : nullptr);
}

// f32x3 and the bf16-store mode: the forward max-pool of a stage boundary runs as an epilogue of the stage's last convolution (epi.h).
// Measured at 854x480 batch 1 (profiles/r03_ab_fusions.txt): saves its launch and 20 us (1.473 -> 1.451 ms of forward convolutions +
// pools).  OSVOS_FUSE_POOL=0 (tests: bit-identity against the separate launches) turns it off.
inline bool fuse_pool(int dtype) {
  static const bool on = [] { const char* e = getenv("OSVOS_FUSE_POOL"); return !(e && e[0] == '0'); }();
  return on && (dtype == OSVOS_F32_X3 || use_store(dtype));
}

// f32x3 stream-K (conv3x3_f32x3.hip): OSVOS_X3_STREAMK = 0 (default) off, 1 the forward's main-stream convolutions, 2 also the data-gradient
// chain.  Built and measured in round 4 (profiles/r04_tune_streamk.txt, docs/DESIGN_rounds_1-4.md 3.9): op level the 64-cout tiles gain 4-7 % (conv1_2, conv4_x)
// and the 128-cout tile loses 1-12 % (conv2_x, conv3_x); in the network, with the automatic choice restricted to the winners, the headline
// loop reads 234.8-235.9 frames/s against 235.2-236.2 without it (three alternating runs on one box) and configs[4] 157.2-157.3 against
// 157.3-157.5 -- the CUs a plain grid leaves without a tile are not idle in the step: the side-branch convolutions of the second stream run
// there.  Mode 2 costs 3.6 % (the weight-gradient stream already fills the data-gradient chain's idle CUs).  Hence opt-in.
inline int streamk_mode() {
  static const int v = [] { const char* e = getenv("OSVOS_X3_STREAMK"); return e ? atoi(e) : 0; }();
  return v;
}

// TIMING ABLATIONS (wrong results; tools/ablate_step.sh) exist only in probe builds (make EXTRA=-DOSVOS_DBG_ABLATIONS): OSVOS_DBG_SKIP bit
// 1 = no slab reduces, 2 = no pool backward kernels, 8 = no side_prep data gradients, 16 = no side_prep weight gradients, 32 = no input
// gradient, 64 = no conv1_1 weight gradient.  The shipped library never skips work.
#ifdef OSVOS_DBG_ABLATIONS
inline int dbg_skip() {
  static const int v = [] { const char* e = getenv("OSVOS_DBG_SKIP"); return e ? atoi(e) : 0; }();
  return v;
}
#else
constexpr int dbg_skip() { return 0; }
#endif

// precision 'fp32x2' (OSVOS_FLAG_X3_TWO_PIECES): the f32x3 kernels of this call take two bf16 pieces per operand (three products) instead of three
// (six); the per-thread switch (errors.cpp) is set for the duration of the call and restored on every return path
extern "C" int osvos_set_x3_pieces(int pieces);
// precision 'fp32h2' (OSVOS_FLAG_X3_HALF_PIECES): two FP16 pieces under block exponents (h2split.h) -- the packs the call reads must have been
// written in that format (osvos_net_pack with the matching flag)
struct PiecesScope {
  int prev;
  explicit PiecesScope(int dtype_) : prev(osvos_x3_pieces()) {
    const bool x3 = (dtype_ & 0xff) == OSVOS_F32_X3;
    osvos_set_x3_pieces(x3 && (dtype_ & OSVOS_FLAG_X3_HALF_PIECES) ? 22 : (x3 && (dtype_ & OSVOS_FLAG_X3_TWO_PIECES) ? 2 : 3));
  }
  ~PiecesScope() { osvos_set_x3_pieces(prev); }
};

// gradient-ready events armed for the next backward of this host thread (osvos_net_arm_grad_events)
struct GradEvents { hipEvent_t ev[OSVOS_NGRAD_GROUPS]; int n = 0; };
GradEvents& grad_events() {
  static thread_local GradEvents g;
  return g;
}

inline double conv_flops(int N, int h, int w, int cin, int cout) { return 2.0 * N * h * w * (double)cout * 9.0 * cin; }

__attribute__((unused)) inline char* at(void* base, size_t off) { return reinterpret_cast<char*>(base) + off; }
inline const char* at(const void* base, size_t off) { return reinterpret_cast<const char*>(base) + off; }

}  // namespace

// device-side helpers implemented in net_kernels.hip
int osvos_gather_small(const float* const* srcs, const size_t* dst_off, const int* counts, int n, void* wbuf, hipStream_t stream);
int osvos_head_grads_finalize(const double* const* part, const int* nblk, const double* fb_part, int fb_nblk,
                              float* const* grads, int accumulate, int have_side, hipStream_t stream);

extern "C" {

size_t osvos_net_wbuf_bytes(int dtype) { return wbuf_layout(dtype & 0xff).total; }
size_t osvos_net_ws_bytes(int N, int H, int W, int dtype) { return ws_layout(N, H, W, dtype & 0xff).total; }
size_t osvos_net_ws_bytes_infer(int N, int H, int W, int dtype) { return ws_layout(N, H, W, dtype & 0xff).fwd_total; }

int osvos_net_ws_query(int N, int H, int W, int dtype, int which, size_t* offset, size_t* elems, int* channels, int* h, int* w) {
  OSVOS_ARG_CHECK(which >= 0 && which <= 21 && offset && elems && channels && h && w, "ws_query: bad arguments");
  WsLayout L = ws_layout(N, H, W, dtype & 0xff);
  ConvDesc d[kNumConv];
  conv_table(d);
  int si, c;
  size_t off;
  if (which < 13) { si = d[which].stage; c = d[which].cout; off = L.act[which]; }
  else if (which < 17) { si = which - 12; c = kStageC[si - 1]; off = L.pooled[si]; }
  else if (which < 21) { si = which - 16; c = 16; off = L.prep[which - 17]; }
  else { si = 0; c = kInPad; off = L.xin; }
  *offset = off; *channels = c; *h = L.hs[si]; *w = L.ws[si];
  *elems = (size_t)N * L.hs[si] * L.ws[si] * c;
  return 0;
}

// storage format of trunk tensor `which` (osvos_net_ws_query 0..16): 0 fp32 NHWC, 1 bf16 NHWC
int osvos_net_ws_format(int dtype, int which) {
  (void)which;
  return use_store(dtype & 0xff) ? 1 : 0;
}

int osvos_net_pack(const float* const* params, void* wbuf, int dtype_, int with_dgrad, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int dtype = dtype_ & 0xff;
  const bool generic = (dtype_ & OSVOS_FLAG_GENERIC_DECONV) != 0;
  OSVOS_ARG_CHECK(params && wbuf, "net_pack: null pointer");
  OSVOS_ARG_CHECK(osvos_dtype_built(dtype), "net_pack: dtype %d not built", dtype);
  for (int i = 0; i < OSVOS_NPARAMS; ++i) OSVOS_ARG_CHECK(params[i] != nullptr, "net_pack: params[%d] is null", i);
  WbufLayout L = wbuf_layout(dtype);
  ConvDesc d[kNumConv];
  conv_table(d);
  const float* srcs[64];
  size_t dsts[64];
  int counts[64];
  int ns = 0;
  // f32x3 with pre-split weights: the fp32 packs are read only where no pre-split pack exists (conv1_1 forward: the exact kernel) and by
  // the input-gradient kernel (layer 0's data-gradient pack); all pre-split packs are formed by ONE launch
  const bool x3ps = dtype == OSVOS_F32_X3 && use_presplit();
  const float* xw[OSVOS_PACK_MAX]; void* xd[OSVOS_PACK_MAX]; int xco[OSVOS_PACK_MAX], xci[OSVOS_PACK_MAX], xdg[OSVOS_PACK_MAX], xhalf[OSVOS_PACK_MAX];
  int nx = 0;
  // precision 'fp32h2': forward and / or data-gradient packs in the FP16-pair format (h2split.h) -- same buffers, other contents
  const int half_fwd = (dtype_ & OSVOS_FLAG_X3_HALF_PIECES) ? 1 : 0, half_bwd = (dtype_ & OSVOS_FLAG_X3_HALF_PIECES_BWD) ? 1 : 0;
  OSVOS_ARG_CHECK(!(half_fwd || half_bwd) || x3ps, "net_pack: the FP16-pair packs exist for dtype OSVOS_F32_X3 with pre-split weights only");
  const bool b16 = dtype == OSVOS_F32_BF16MFMA;      // all bf16 packs (17 filters x 2 forms) in ONE launch too (round 5 prep)
  for (int l = 0; l < kNumConv; ++l) {
    int rc;
    if (b16) {
      xw[nx] = params[d[l].w_param]; xd[nx] = at(wbuf, L.fwd[l]); xco[nx] = d[l].cout; xci[nx] = d[l].cin; xdg[nx] = 0; ++nx;
      if (with_dgrad) { xw[nx] = params[d[l].w_param]; xd[nx] = at(wbuf, L.dgrad[l]); xco[nx] = d[l].cout; xci[nx] = d[l].cin; xdg[nx] = 1; ++nx; }
      srcs[ns] = params[d[l].b_param]; dsts[ns] = L.bias[l]; counts[ns] = d[l].cout; ++ns;
      continue;
    }
    if (!(x3ps && L.fwd3[l] != (size_t)-1) && (rc = osvos_pack_conv3x3_fwd(params[d[l].w_param], at(wbuf, L.fwd[l]), d[l].cout, d[l].cin, dtype, stream))) return rc;
    if (with_dgrad && !(x3ps && L.dgrad3[l] != (size_t)-1 && l != 0) &&
        (rc = osvos_pack_conv3x3_dgrad(params[d[l].w_param], at(wbuf, L.dgrad[l]), d[l].cout, d[l].cin, dtype, stream))) return rc;
    if (L.fwd3[l] != (size_t)-1) { xw[nx] = params[d[l].w_param]; xd[nx] = at(wbuf, L.fwd3[l]); xco[nx] = d[l].cout; xci[nx] = d[l].cin; xdg[nx] = 0; xhalf[nx] = half_fwd; ++nx; }
    if (with_dgrad && L.dgrad3[l] != (size_t)-1) { xw[nx] = params[d[l].w_param]; xd[nx] = at(wbuf, L.dgrad3[l]); xco[nx] = d[l].cout; xci[nx] = d[l].cin; xdg[nx] = 1; xhalf[nx] = half_bwd; ++nx; }
    srcs[ns] = params[d[l].b_param]; dsts[ns] = L.bias[l]; counts[ns] = d[l].cout; ++ns;
  }
  if (nx > 0) {
    const int rc = b16 ? osvos_pack_bf16_multi(xw, xd, xco, xci, xdg, nx, stream) : osvos_pack_x3_multi_fmt(xw, xd, xco, xci, xdg, xhalf, nx, stream);
    if (rc) return rc;
  }

  for (int i = 0; i < 4; ++i) {
    const int k = 4 << i;
    srcs[ns] = params[42 + 2 * i]; dsts[ns] = L.wd[i]; counts[ns] = 16; ++ns;
    srcs[ns] = params[43 + 2 * i]; dsts[ns] = L.bd[i]; counts[ns] = 1; ++ns;
    srcs[ns] = params[4 + i]; dsts[ns] = L.f1[i]; counts[ns] = k * k; ++ns;     // upscale_[i].weight[0,0]
    srcs[ns] = params[i]; dsts[ns] = L.f16[i]; counts[ns] = k * k; ++ns;         // upscale[i].weight[0,0]
  }
  srcs[ns] = params[50]; dsts[ns] = L.wf; counts[ns] = 64; ++ns;
  srcs[ns] = params[51]; dsts[ns] = L.bf; counts[ns] = 1; ++ns;
  int rc = osvos_gather_small(srcs, dsts, counts, ns, wbuf, stream);
  if (rc || !generic) return rc;
  for (int i = 0; i < 4; ++i) {      // Weff_i = sum_co wfuse[16 i + co] * upscale[i].weight[:, co]
    rc = osvos_head_weff(params[i], params[50] + 16 * i, reinterpret_cast<float*>(at(wbuf, L.weff[i])), 4 << i, stream);
    if (rc) return rc;
    const int k = 4 << i;
    OSVOS_HIP_CHECK(hipMemcpyAsync(at(wbuf, L.wup[i]), params[i], sizeof(float) * 256 * k * k, hipMemcpyDeviceToDevice, stream));
  }
  return 0;
}

int osvos_net_forward(const float* x_nchw, const void* wbuf, void* ws, float* const* outs,
                      int N, int H, int W, int dtype_, void* stream_, void* aux_stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int dtype = dtype_ & 0xff;
  const bool generic = (dtype_ & OSVOS_FLAG_GENERIC_DECONV) != 0;
  const bool infer = (dtype_ & OSVOS_FLAG_INFERENCE) != 0;      // no backward will read the sign bits / pool codes: do not write them
  PiecesScope pieces_scope(dtype_);
  hipStream_t aux_all = aux_stream_ ? (hipStream_t)aux_stream_ : stream;
  const bool two = aux_all != stream;
  EventPool& evp = event_pool();
  OSVOS_ARG_CHECK(x_nchw && wbuf && ws && outs, "net_forward: null pointer");
  OSVOS_ARG_CHECK(osvos_dtype_built(dtype), "net_forward: dtype %d not built", dtype);
  OSVOS_ARG_CHECK(N > 0 && H > 0 && W > 0, "net_forward: bad shape %dx%dx%d", N, H, W);
  for (int i = 0; i < 5; ++i) OSVOS_ARG_CHECK(outs[i] != nullptr, "net_forward: outs[%d] is null", i);
  const WbufLayout P = wbuf_layout(dtype);
  const WsLayout L = ws_layout(N, H, W, dtype);
  ConvDesc d[kNumConv];
  conv_table(d);
  const bool store = use_store(dtype);
  auto sh = [&](size_t off) -> void* { return store ? at(ws, off) : nullptr; };      // bf16 tensor (copy, or the only one)
  auto f32 = [&](size_t off) -> void* { return store ? nullptr : at(ws, off); };                  // fp32 trunk tensor (absent in store mode)
  int rc = osvos_nchw_to_nhwc_f32(x_nchw, reinterpret_cast<float*>(at(ws, L.xin)), sh(L.xin_b), N, 3, H, W, kInPad, stream);
  if (rc) return rc;
  // stream-K workspace of the main-stream convolutions: the tickets must be zero before the first launch (every launch leaves them zero; the
  // workspace itself arrives uninitialised from the caller, so they are cleared once per forward -- a 32 KB memset node, capturable)
  void* const sk_ws = (L.sk_ws != (size_t)-1 && streamk_mode() >= 1) ? at(ws, L.sk_ws) : nullptr;
  if (sk_ws != nullptr) OSVOS_HIP_CHECK(hipMemsetAsync(sk_ws, 0, osvos_conv3x3_f32x3_streamk_ticket_bytes(), stream));
  const void* cur = at(ws, L.xin);
  const void* cur_b = sh(L.xin_b);
  int l = 0;
  const float* score[4]; const float* fpart[4]; const float* f1[4]; const float* f16[4];
  for (int si = 0; si < 5; ++si) {
    const int h = L.hs[si], w = L.ws[si];
    if (si > 0 && fuse_pool(dtype)) {      // pooled[si] was written by the previous stage's last convolution
      cur = at(ws, L.pooled[si]);
      cur_b = store ? at(ws, L.pooled_b[si]) : nullptr;
    } else if (si > 0) {
      if (store)
        rc = osvos_maxpool2x2_bf16_code(cur_b, at(ws, L.pooled_b[si]), (L.pool_code[si] != (size_t)-1 && !infer) ? at(ws, L.pool_code[si]) : nullptr, N, L.hs[si - 1],
                                        L.ws[si - 1], kStageC[si - 1], stream);
      else
        rc = osvos_maxpool2x2_f32(reinterpret_cast<const float*>(cur), reinterpret_cast<float*>(at(ws, L.pooled[si])), sh(L.pooled_b[si]), N,
                                  L.hs[si - 1], L.ws[si - 1], kStageC[si - 1], stream);
      if (rc) return rc;
      cur = at(ws, L.pooled[si]);
      cur_b = sh(L.pooled_b[si]);
    }
    for (int j = 0; j < kStageN[si]; ++j, ++l) {
      {
        ProfScope ps(OSVOS_PROF_CONV_FWD, conv_flops(N, h, w, d[l].cin, d[l].cout), stream);
        ConvEpi epi;
        const bool pool_here = fuse_pool(dtype) && si < 4 && j == kStageN[si] - 1;      // last convolution of stages 0-3: + the pooled tensor
        if (pool_here && !store) epi.pooled = reinterpret_cast<float*>(at(ws, L.pooled[si + 1]));
        rc = conv_main(cur, cur_b, at(wbuf, P.fwd[l]), reinterpret_cast<const float*>(at(wbuf, P.bias[l])), nullptr, nullptr,
                       f32(L.act[l]), sh(L.act_b[l]), N, h, w, d[l].cin_s, d[l].cout, d[l].cout, 1, dtype, at(ws, L.conv_part), stream,
                       P.fwd3[l] != (size_t)-1 ? at(wbuf, P.fwd3[l]) : nullptr, (pool_here && !store) ? &epi : nullptr, nullptr,
                       (L.bits[l] != (size_t)-1 && !infer) ? at(ws, L.bits[l]) : nullptr, (pool_here && store) ? at(ws, L.pooled_b[si + 1]) : nullptr, sk_ws,
                       (pool_here && store && L.pool_code[si + 1] != (size_t)-1 && !infer) ? at(ws, L.pool_code[si + 1]) : nullptr);
      }
      if (rc) return rc;
      cur = at(ws, L.act[l]);
      cur_b = sh(L.act_b[l]);
    }
    if (si > 0) {
      // side branch of this stage (skinny Cout=16 conv + the two 1x1 dots): latency-bound launches
      // that run on the aux stream in the shadow of the next stage's big convolutions
      const int i = si - 1, sl = kNumTrunk + i;
      // the LAST stage's side branch has nothing left to hide behind: on the main stream it saves two cross-stream event hops (~12-25 us each)
      hipStream_t aux = (si == 4) ? stream : aux_all;
      if (two && aux != stream) {
        hipEvent_t e = evp.next();
        if (!e) return -1;
        OSVOS_HIP_CHECK(hipEventRecord(e, stream));
        OSVOS_HIP_CHECK(hipStreamWaitEvent(aux, e, 0));
      }
      {
        ProfScope ps(OSVOS_PROF_OTHER, conv_flops(N, h, w, d[sl].cin, 16), aux);
        rc = conv_main(cur, cur_b, at(wbuf, P.fwd[sl]), reinterpret_cast<const float*>(at(wbuf, P.bias[sl])), nullptr, nullptr,
                       at(ws, L.prep[i]), nullptr, N, h, w, d[sl].cin_s, 16, 16, 0, dtype, L.side_part[i] != (size_t)-1 ? at(ws, L.side_part[i]) : nullptr, aux,
                       P.fwd3[sl] != (size_t)-1 ? at(wbuf, P.fwd3[sl]) : nullptr);
      }
      if (rc) return rc;
      float* sc = reinterpret_cast<float*>(at(ws, L.score[i]));
      float* fp = reinterpret_cast<float*>(at(ws, L.fpart[i]));
      rc = osvos_head_lowres(at(ws, L.prep[i]), reinterpret_cast<const float*>(at(wbuf, P.wd[i])),
                             reinterpret_cast<const float*>(at(wbuf, P.bd[i])),
                             reinterpret_cast<const float*>(at(wbuf, P.wf)) + 16 * i, sc, fp, N, h, w, dtype, aux);
      if (rc) return rc;
      score[i] = sc; fpart[i] = fp;
      f1[i] = reinterpret_cast<const float*>(at(wbuf, P.f1[i]));
      f16[i] = reinterpret_cast<const float*>(at(wbuf, P.f16[i]));
    }
  }
  if (two) {
    hipEvent_t e = evp.next();
    if (!e) return -1;
    OSVOS_HIP_CHECK(hipEventRecord(e, aux_all));
    OSVOS_HIP_CHECK(hipStreamWaitEvent(stream, e, 0));
  }
  if (generic) {      // non-diagonal upscale weights: fused head from the 16-channel side_prep outputs and Weff (head_generic.hip)
    const float* prep[4]; const float* weff[4];
    for (int i = 0; i < 4; ++i) { prep[i] = reinterpret_cast<const float*>(at(ws, L.prep[i])); weff[i] = reinterpret_cast<const float*>(at(wbuf, P.weff[i])); }
    return osvos_head_upsample_generic(score, prep, f1, weff, reinterpret_cast<const float*>(at(wbuf, P.bf)), outs, N, H, W, &L.hs[1], &L.ws[1], stream);
  }
  return osvos_head_upsample(score, fpart, f1, f16, reinterpret_cast<const float*>(at(wbuf, P.bf)), outs, N, H, W,
                             &L.hs[1], &L.ws[1], stream);
}

int osvos_net_join(void* stream_, void* aux_, void* aux2_) {
  hipStream_t stream = (hipStream_t)stream_;
  EventPool& evp = event_pool();
  for (void* a : {aux_, aux2_}) {
    if (a == nullptr || a == stream_) continue;
    hipEvent_t e = evp.next();
    if (!e) return -1;
    OSVOS_HIP_CHECK(hipEventRecord(e, (hipStream_t)a));
    OSVOS_HIP_CHECK(hipStreamWaitEvent(stream, e, 0));
  }
  return 0;
}

int osvos_net_arm_grad_events(void* const* events, int n) {
  OSVOS_ARG_CHECK(n >= 0 && n <= OSVOS_NGRAD_GROUPS && (n == 0 || events != nullptr), "arm_grad_events: n = %d (0..%d)", n, OSVOS_NGRAD_GROUPS);
  GradEvents& g = grad_events();
  g.n = n;
  for (int k = 0; k < n; ++k) g.ev[k] = (hipEvent_t)events[k];
  return 0;
}

int osvos_net_backward(const void* wbuf, void* ws, const float* const* douts, float* const* grads,
                       float* dx_nchw, int N, int H, int W, int dtype_, int accumulate, void* stream_, void* aux_stream_,
                       void* aux2_stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int dtype = dtype_ & 0xff;
  const bool generic = (dtype_ & OSVOS_FLAG_GENERIC_DECONV) != 0;
  PiecesScope pieces_scope(dtype_);
  hipStream_t aux = aux_stream_ ? (hipStream_t)aux_stream_ : stream;
  hipStream_t aux2 = aux2_stream_ ? (hipStream_t)aux2_stream_ : aux;
  const GradEvents gev = grad_events();      // armed for this call only
  grad_events().n = 0;
  // deferred join (OSVOS_FLAG_DEFER_JOIN): not with armed gradient-ready events (their last group is recorded behind the join)
  const bool defer_join = (dtype_ & OSVOS_FLAG_DEFER_JOIN) != 0 && gev.n == 0 && aux != stream;
  auto ready = [&](int group, hipStream_t st) -> int {      // group's gradients are complete once `st` gets here
    if (group < gev.n && gev.ev[group] != nullptr) OSVOS_HIP_CHECK(hipEventRecord(gev.ev[group], st));
    return 0;
  };
  const bool two = aux != stream;
  const bool three = aux2 != aux;
  OSVOS_ARG_CHECK(wbuf && ws && douts && grads, "net_backward: null pointer");
  OSVOS_ARG_CHECK(osvos_dtype_built(dtype), "net_backward: dtype %d not built", dtype);
  const WbufLayout P = wbuf_layout(dtype);
  const WsLayout L = ws_layout(N, H, W, dtype);
  ConvDesc d[kNumConv];
  conv_table(d);
  const bool store = use_store(dtype);
  auto sh = [&](size_t off) -> void* { return store ? at(ws, off) : nullptr; };
  auto f32 = [&](size_t off) -> void* { return store ? nullptr : at(ws, off); };
  auto mk32 = [&](size_t off) -> const void* { return store ? nullptr : at(ws, off); };           // ReLU mask operand: fp32 ...
  auto mk16 = [&](size_t off) -> const void* { return store ? at(ws, off) : nullptr; };           // ... or bf16
  // fork: aux waits for everything enqueued on `stream` so far; join: `stream` waits for aux.
  // The weight-gradient kernels run on aux concurrently with the data-gradient kernel of the same
  // layer: both are MFMA kernels with independent stalls (barriers, LDS latency, tails), and
  // together they keep the matrix pipes busier than either does alone.
  EventPool& evp = event_pool();
  auto join = [&]() -> int {
    if (defer_join) return 0;            // the caller joins (osvos_net_join) before it touches a parameter gradient
    if (two) {
      hipEvent_t e = evp.next();
      if (!e) return -1;
      OSVOS_HIP_CHECK(hipEventRecord(e, aux));
      OSVOS_HIP_CHECK(hipStreamWaitEvent(stream, e, 0));
    }
    if (three) {
      hipEvent_t e = evp.next();
      if (!e) return -1;
      OSVOS_HIP_CHECK(hipEventRecord(e, aux2));
      OSVOS_HIP_CHECK(hipStreamWaitEvent(stream, e, 0));
    }
    return 0;
  };
  // weight gradient of one layer: partial slabs on aux (MFMA kernel), slab reduce on aux2 (bandwidth kernel)
  // (store mode: the wide layers read both operands as bf16; conv1_1 and side_prep keep their exact-fp32 skinny kernels, fed
  //  with the bf16 tensor on the wide side -- xin stays fp32 for conv1_1, dprep for side_prep)
  auto wgrad_launch = [&](const void* xin, const void* g, int l, int h, int w, hipStream_t st) -> int {
    if (store) {
      if (osvos_wgrad_bf16_applicable(d[l].cin_s, d[l].cout) && d[l].cin == d[l].cin_s)
        return osvos_conv3x3_wgrad_bf16mfma_io(xin, g, 1, at(ws, L.wgrad[l]), grads[d[l].w_param], grads[d[l].b_param], N, h, w,
                                               d[l].cin, d[l].cin_s, d[l].cout, d[l].cout, accumulate, st);
      const int r = osvos_conv3x3_wgrad_small_f32(xin, g, 1, at(ws, L.wgrad[l]), grads[d[l].w_param], grads[d[l].b_param], N, h, w,
                                                  d[l].cin, d[l].cin_s, d[l].cout, d[l].cout, accumulate, st);
      if (r == 1) osvos_set_error("net_backward: no bf16-store weight-gradient kernel for layer %d", l);
      return r;
    }
    return osvos_conv3x3_wgrad(xin, g, at(ws, L.wgrad[l]), grads[d[l].w_param], grads[d[l].b_param], N, h, w,
                               d[l].cin, d[l].cin_s, d[l].cout, d[l].cout, accumulate, dtype, st);
  };
  auto wgrad = [&](const void* xin, const void* g, int l, int h, int w) -> int {
    int r;
    if (!three) return wgrad_launch(xin, g, l, h, w, aux);
    osvos_wgrad_set_phase(1);
    r = wgrad_launch(xin, g, l, h, w, aux);
    osvos_wgrad_set_phase(0);
    if (r) return r;
    if (dbg_skip() & 1) return 0;
    hipEvent_t e = evp.next();
    if (!e) return -1;
    OSVOS_HIP_CHECK(hipEventRecord(e, aux));
    OSVOS_HIP_CHECK(hipStreamWaitEvent(aux2, e, 0));
    osvos_wgrad_set_phase(2);
    r = wgrad_launch(xin, g, l, h, w, aux2);
    osvos_wgrad_set_phase(0);
    return r;
  };
  int rc;
  double* acc = reinterpret_cast<double*>(at(ws, L.acc));
  const double* part[4];
  int nblk[4], fb_nblk = 0;
  bool have_side = false;
  for (int i = 0; i < 4; ++i) have_side = have_side || douts[i] != nullptr;
  const float* dfused = douts[4];

  // ---- head: upstream full-resolution gradients -> dprep[i] (+ score_dsn / fuse gradients) ----
  if (!generic) {        // the four scales in one launch (as four ~20 us launches they sit back to back on the critical path)
    const float *prep4[4], *f1_4[4], *f16_4[4], *wd4[4];
    float* dprep4[4];
    void* dprepb4[4];
    double* acc4[4];
    for (int i = 0; i < 4; ++i) {
      prep4[i] = reinterpret_cast<const float*>(at(ws, L.prep[i]));
      f1_4[i] = reinterpret_cast<const float*>(at(wbuf, P.f1[i]));
      f16_4[i] = reinterpret_cast<const float*>(at(wbuf, P.f16[i]));
      wd4[i] = reinterpret_cast<const float*>(at(wbuf, P.wd[i]));
      dprep4[i] = reinterpret_cast<float*>(at(ws, L.dprep[i]));
      dprepb4[i] = store ? at(ws, L.dprep_b[i]) : nullptr;
      acc4[i] = acc + (size_t)i * OSVOS_HEAD_MAX_BLOCKS * 34;
      part[i] = acc4[i];
      nblk[i] = osvos_head_bwd_blocks(N, L.hs[i + 1], L.ws[i + 1], i);
    }
    rc = osvos_head_bwd4_f32(prep4, douts, dfused, f1_4, f16_4, wd4, reinterpret_cast<const float*>(at(wbuf, P.wf)), dprep4, dprepb4, acc4, N, H, W,
                             &L.hs[1], &L.ws[1], stream);
    if (rc) return rc;
  } else
  for (int i = 0; i < 4; ++i) {
    const int si = i + 1;
    if (generic)
      rc = osvos_head_bwd_generic(reinterpret_cast<const float*>(at(ws, L.prep[i])), douts[i], dfused, reinterpret_cast<const float*>(at(wbuf, P.f1[i])),
                                  reinterpret_cast<const float*>(at(wbuf, P.weff[i])), reinterpret_cast<const float*>(at(wbuf, P.wd[i])),
                                  reinterpret_cast<float*>(at(ws, L.dprep[i])), store ? at(ws, L.dprep_b[i]) : nullptr, acc + (size_t)i * OSVOS_HEAD_MAX_BLOCKS * 34,
                                  N, H, W, L.hs[si], L.ws[si], i, stream);
    else
      rc = osvos_head_bwd_f32(reinterpret_cast<const float*>(at(ws, L.prep[i])), douts[i], dfused, reinterpret_cast<const float*>(at(wbuf, P.f1[i])),
                        reinterpret_cast<const float*>(at(wbuf, P.f16[i])), reinterpret_cast<const float*>(at(wbuf, P.wd[i])),
                        reinterpret_cast<const float*>(at(wbuf, P.wf)) + 16 * i, reinterpret_cast<float*>(at(ws, L.dprep[i])),
                        store ? at(ws, L.dprep_b[i]) : nullptr, acc + (size_t)i * OSVOS_HEAD_MAX_BLOCKS * 34,
                        N, H, W, L.hs[si], L.ws[si], i, stream);
    if (rc) return rc;
    part[i] = acc + (size_t)i * OSVOS_HEAD_MAX_BLOCKS * 34;
    nblk[i] = osvos_head_bwd_blocks(N, L.hs[si], L.ws[si], i);
  }
  double* fb_part = acc + (size_t)4 * OSVOS_HEAD_MAX_BLOCKS * 34;
  if (dfused != nullptr) {
    rc = osvos_sum_partials(dfused, (long)N * H * W, fb_part, &fb_nblk, stream);
    if (rc) return rc;
  }
  // The finalize kernel only feeds PARAMETER gradients (score_dsn, fuse): off the critical path -- on the reduce stream (or the weight-gradient
  // stream) behind an event, so that the data-gradient chain starts one launch earlier (round 6; workspace partials only: nothing of the caller's
  // is read off `stream`).  The generic head's extra reductions stay on `stream`.
  OSVOS_ENV_INT(fin_side, "OSVOS_FINALIZE_SIDE", 1);      // A/B switch (0: on `stream`, as rounds 1-5)
  hipStream_t fin = (!generic && two && fin_side) ? aux2 : stream;
  if (fin != stream) {
    hipEvent_t e = evp.next();
    if (!e) return -1;
    OSVOS_HIP_CHECK(hipEventRecord(e, stream));
    OSVOS_HIP_CHECK(hipStreamWaitEvent(fin, e, 0));
  }
  rc = osvos_head_grads_finalize(part, nblk, fb_part, fb_nblk, grads, accumulate, have_side ? 1 : 0, fin);
  if (rc) return rc;
  if (generic) {
    // fuse.weight / upscale[i].weight gradients from the tap-indexed sums G_i (the partials above carried zeros for fuse.weight);
    // upscale_[i].weight (the 1 -> 1 side deconv) likewise when asked for.  The upscale weights themselves are read from the copy
    // osvos_net_pack left in wbuf.
    for (int i = 0; i < 4; ++i) {
      const int si = i + 1, k = 4 << i;
      double* G = reinterpret_cast<double*>(at(ws, L.gbuf[i]));
      if (dfused != nullptr && (grads[50] != nullptr || grads[i] != nullptr)) {
        rc = osvos_head_tapsum(reinterpret_cast<const float*>(at(ws, L.prep[i])), 16, dfused, G, N, H, W, L.hs[si], L.ws[si], i, stream);
        if (rc) return rc;
        rc = osvos_head_generic_param_grads(reinterpret_cast<const float*>(at(wbuf, P.wup[i])), reinterpret_cast<const float*>(at(wbuf, P.wf)) + 16 * i, G,
                                            grads[50] ? grads[50] + 16 * i : nullptr, grads[i], k, accumulate, stream);
        if (rc) return rc;
      } else if (grads[i] != nullptr && !accumulate) {
        OSVOS_HIP_CHECK(hipMemsetAsync(grads[i], 0, sizeof(float) * 256 * k * k, stream));
      }
      if (grads[4 + i] != nullptr) {
        if (douts[i] != nullptr) {
          rc = osvos_head_tapsum(reinterpret_cast<const float*>(at(ws, L.score[i])), 1, douts[i], G + 16 * k * k, N, H, W, L.hs[si], L.ws[si], i, stream);
          if (rc) return rc;
          rc = osvos_head_dw1(G + 16 * k * k, grads[4 + i], k, accumulate, stream);
          if (rc) return rc;
        } else if (!accumulate) {
          OSVOS_HIP_CHECK(hipMemsetAsync(grads[4 + i], 0, sizeof(float) * k * k, stream));
        }
      }
    }
  }
  if ((rc = ready(0, fin))) return rc;      // score_dsn + fuse gradients

  // ---- data-gradient chain on `stream`, weight gradients trailing on `aux` ----------------------
  // ready[k]: event recorded on `stream` when the k-th upstream gradient tensor is complete
  double bwd_flops = 0.0;
  for (int l = 0; l < kNumConv; ++l) bwd_flops += 2.0 * conv_flops(N, L.hs[d[l].stage], L.ws[d[l].stage], d[l].cin, d[l].cout);
  if (dx_nchw == nullptr) bwd_flops -= conv_flops(N, H, W, 3, d[0].cout);
  ProfScope ps(OSVOS_PROF_CONV_BWD, bwd_flops, stream);
  auto signal = [&]() -> int {          // aux may consume everything `stream` has produced so far
    if (!two) return 0;
    hipEvent_t e = evp.next();
    if (!e) return -1;
    OSVOS_HIP_CHECK(hipEventRecord(e, stream));
    OSVOS_HIP_CHECK(hipStreamWaitEvent(aux, e, 0));
    return 0;
  };
  if ((rc = signal())) return rc;       // dprep[0..3] ready
  // the four skinny side_prep weight gradients head the weight-gradient stream (on the third stream beside the first trunk gradients they
  // measured +0.1-0.25 %, inside the noise: round 3)
  for (int i = 0; i < 4; ++i) {
    const int si = i + 1, sl = kNumTrunk + i, h = L.hs[si], w = L.ws[si];
    const int lx = last_of_stage(si);
    if (grads[d[sl].w_param] != nullptr && !(dbg_skip() & 16)) {
      const void* dp = store ? at(ws, L.dprep_b[i]) : at(ws, L.dprep[i]);      // (store mode: bf16 x and bf16 dprep)
      rc = wgrad(at(ws, L.act[lx]), dp, sl, h, w);
      if (rc) return rc;
    }
  }
  if ((rc = ready(1, aux2))) return rc;        // the four side_prep layers (their slab reduces are the last writers, in order, on aux2)
  // The side branches' data gradients (16 -> C channels: one K chunk, all prologue and epilogue, 13-51 us each at batch 1).  Only stage 4's is needed
  // at once (it IS the upstream gradient of conv5_3); dside of stages 3, 2, 1 is consumed three, six and nine trunk layers later, by the pooling
  // backward of the stage boundary.  OSVOS_SIDE_DGRAD_ASIDE=1 runs those three on the reduce stream beside the head of the chain instead of in
  // front of it.  Measured in round 6 and NOT the default: the backward is bound by the chip's throughput over BOTH chains (the weight-gradient
  // stream ends within 3 us of the data-gradient chain), not by the chain's length -- 238.2 vs 237.4 frames/s at batch 1 (noise), 1147 vs 1157 on
  // configs[2] and 300 vs 306 with the two-piece backward (both slightly worse): same-box alternating rounds, tools/ab_env.sh.
  OSVOS_ENV_INT(side_aside, "OSVOS_SIDE_DGRAD_ASIDE", 0);
  hipStream_t sds = (two && side_aside) ? aux2 : stream;
  hipEvent_t side_ev[3] = {nullptr, nullptr, nullptr};
  if (sds != stream) {
    hipEvent_t e = evp.next();
    if (!e) return -1;
    OSVOS_HIP_CHECK(hipEventRecord(e, stream));      // dprep[0..3] ready
    OSVOS_HIP_CHECK(hipStreamWaitEvent(sds, e, 0));
  }
  for (int i = 3; i >= 0; --i) {
    const int si = i + 1, sl = kNumTrunk + i, h = L.hs[si], w = L.ws[si];
    const int lx = last_of_stage(si);
    // stage 4 has no pool after it: its ReLU mask is applied right here and the result is the
    // upstream gradient of conv5_3; stages 1-3 are merged in maxpool2x2_bwd below
    void* dst = (i == 3) ? f32(L.dy[lx]) : f32(L.dside[i]);
    void* dst_b = (i == 3) ? sh(L.dy_b[lx]) : (store ? at(ws, L.dside_b[i]) : nullptr);
    if (dbg_skip() & 8) continue;
    hipStream_t st = (i == 3) ? stream : sds;
    rc = conv_main(at(ws, L.dprep[i]), store ? at(ws, L.dprep_b[i]) : nullptr, at(wbuf, P.dgrad[sl]), nullptr, (i == 3) ? mk32(L.act[lx]) : nullptr,
                   (i == 3) ? mk16(L.act_b[lx]) : nullptr, dst, dst_b, N, h, w, 16, d[sl].cin, d[sl].cin, 0, dtype, nullptr, st,
                   P.dgrad3[sl] != (size_t)-1 ? at(wbuf, P.dgrad3[sl]) : nullptr, nullptr,
                   (i == 3 && L.bits[lx] != (size_t)-1) ? at(ws, L.bits[lx]) : nullptr);
    if (rc) return rc;
    if (st != stream) {
      side_ev[i] = evp.next();
      if (!side_ev[i]) return -1;
      OSVOS_HIP_CHECK(hipEventRecord(side_ev[i], st));
    }
  }

  // trunk, deepest layer first; dy[l] = dLoss/d(conv l output), ReLU mask already applied
  // (stream-K for the data-gradient chain only with OSVOS_X3_STREAMK=2; the workspace's tickets are zero: every launch leaves them so)
  void* const sk_bwd = (L.sk_ws != (size_t)-1 && streamk_mode() >= 2) ? at(ws, L.sk_ws) : nullptr;
  for (int l = kNumTrunk - 1; l >= 0; --l) {
    const int si = d[l].stage, h = L.hs[si], w = L.ws[si];
    const bool first_of_stage = (l == 0) || d[l - 1].stage != si;
    const void* xin = first_of_stage ? (si == 0 ? at(ws, L.xin) : at(ws, L.pooled[si])) : at(ws, L.act[l - 1]);
    const void* g = at(ws, L.dy[l]);
    const void* g_b = sh(L.dy_b[l]);
    // (Measured twice in round 3: conv1_1's weight gradient on the third stream BESIDE the input gradient instead of behind it is neutral at
    //  batch 1 (f32x3) and 0.5-1 % slower at batch 12 (bf16) -- not built in.)
    // conv1_1's weight gradient is the LAST piece of work of the step and the weight-gradient stream is the one that finishes last (conv1_2's
    // gradient is still running when the data-gradient chain ends): it goes on the main stream, behind the input gradient, beside conv1_2's
    const bool tail_on_main = l == 0 && two && !defer_join;      // (deferred join: everything gradient-related stays on the side streams)
    if (grads[d[l].w_param] != nullptr && !tail_on_main) {
      if ((rc = signal())) return rc;   // dy[l] ready -> its weight gradient may start on aux
      rc = wgrad(xin, g, l, h, w);
      if (rc) return rc;
    }
    if (first_of_stage && !tail_on_main && (rc = ready(2 + (4 - si), aux2))) return rc;      // stage si complete (its first conv is the last one processed)
    if (l == 0) {
      if (dbg_skip() & 32) {
      } else if (dx_nchw != nullptr && (dtype == OSVOS_F32 || dtype == OSVOS_F32_X3)) {
        rc = osvos_conv3x3_dgrad_c3_f32(reinterpret_cast<const float*>(g), reinterpret_cast<const float*>(at(wbuf, P.dgrad[0])), dx_nchw, N, h, w, d[0].cout, stream);
        if (rc) return rc;
      } else if (dx_nchw != nullptr && g_b != nullptr && d[0].cout == 64) {
        // bf16 trunk tensors: the whole 64-channel halo tile once through LDS, filter rows on the matrix pipe, straight into NCHW (dgrad_c3.hip;
        // same-box A/B at batch 12, two alternating rounds: 1118.7 / 1114.1 frames/s against 1099.1 / 1106.4 with the 32-cout tile + layout
        // kernel below; the fp32-FMA kernel fed with bf16, a first attempt of the round, read 1073-1081 and is gone)
        rc = osvos_conv3x3_dgrad_c3_bf16mfma(g_b, at(wbuf, P.dgrad[0]), dx_nchw, N, h, w, d[0].cout, stream);
        if (rc) return rc;
      } else if (dx_nchw != nullptr) {
        rc = conv_main(g, g_b, at(wbuf, P.dgrad[0]), nullptr, nullptr, nullptr, at(ws, L.dxin), nullptr, N, h, w, d[0].cout, 3, 4, 0, dtype,
                       nullptr, stream, P.dgrad3[0] != (size_t)-1 ? at(wbuf, P.dgrad3[0]) : nullptr);
        if (rc) return rc;
        rc = osvos_nhwc_to_nchw(at(ws, L.dxin), dx_nchw, N, 3, H, W, 4, dtype, stream);
        if (rc) return rc;
      }
      if (tail_on_main) {
        if (grads[d[0].w_param] != nullptr && !(dbg_skip() & 64) && (rc = wgrad_launch(xin, g, 0, h, w, stream))) return rc;
        if ((rc = join())) return rc;
        return ready(2 + 4, stream);      // stage 0 complete: conv1_2's reduce (aux2, joined) and conv1_1's (main)
      }
      break;
    }
    if (first_of_stage) {
      // through the pool into the previous stage's output (+ that stage's side branch, + ReLU mask)
      rc = conv_main(g, g_b, at(wbuf, P.dgrad[l]), nullptr, nullptr, nullptr, f32(L.dpool[si]), store ? at(ws, L.dpool_b[si]) : nullptr, N, h, w,
                     d[l].cout, d[l].cin, d[l].cin, 0, dtype, at(ws, L.conv_part), stream, P.dgrad3[l] != (size_t)-1 ? at(wbuf, P.dgrad3[l]) : nullptr,
                     nullptr, nullptr, nullptr, nullptr, sk_bwd);
      if (rc) return rc;
      const int ps2 = si - 1;
      const void* dside = ps2 >= 1 ? at(ws, L.dside[ps2 - 1]) : nullptr;
      if (dbg_skip() & 2) continue;
      if (ps2 >= 1 && side_ev[ps2 - 1] != nullptr) OSVOS_HIP_CHECK(hipStreamWaitEvent(stream, side_ev[ps2 - 1], 0));      // dside written on the reduce stream
      if (store && L.pool_code[si] != (size_t)-1)      // one code byte per pooled element instead of the pool's four inputs
        rc = osvos_maxpool2x2_bwd_bf16_code(at(ws, L.pool_code[si]), at(ws, L.dpool_b[si]), dside, at(ws, L.dy_b[l - 1]), N, L.hs[ps2], L.ws[ps2],
                                            kStageC[ps2], stream);
      else if (store)
        rc = osvos_maxpool2x2_bwd_bf16(at(ws, L.act_b[l - 1]), at(ws, L.dpool_b[si]), dside, at(ws, L.dy_b[l - 1]), N, L.hs[ps2], L.ws[ps2],
                                       kStageC[ps2], stream);
      else
        rc = osvos_maxpool2x2_bwd_f32(reinterpret_cast<const float*>(at(ws, L.act[l - 1])), reinterpret_cast<const float*>(at(ws, L.dpool[si])),
                                      reinterpret_cast<const float*>(dside), reinterpret_cast<float*>(at(ws, L.dy[l - 1])), sh(L.dy_b[l - 1]), N,
                                      L.hs[ps2], L.ws[ps2], kStageC[ps2], stream);
      if (rc) return rc;
    } else {
      rc = conv_main(g, g_b, at(wbuf, P.dgrad[l]), nullptr, mk32(L.act[l - 1]), mk16(L.act_b[l - 1]), f32(L.dy[l - 1]), sh(L.dy_b[l - 1]), N, h, w,
                     d[l].cout, d[l].cin, d[l].cin, 0, dtype, at(ws, L.conv_part), stream, P.dgrad3[l] != (size_t)-1 ? at(wbuf, P.dgrad3[l]) : nullptr,
                     nullptr, L.bits[l - 1] != (size_t)-1 ? at(ws, L.bits[l - 1]) : nullptr, nullptr, nullptr, sk_bwd);
      if (rc) return rc;
    }
  }
  if ((rc = join())) return rc;         // everything is back on `stream` when the call returns
  return 0;
}

}  // extern "C"
