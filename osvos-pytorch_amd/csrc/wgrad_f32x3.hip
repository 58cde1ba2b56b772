// 3x3 convolution weight gradient, fp32 tensors in / fp32 gradient out, on the bf16 matrix pipe with THREE-WAY SPLIT operands
// (the weight-gradient third of dtype OSVOS_F32_X3; see conv3x3_f32x3.hip for the arithmetic: v = hi + mid + lo exactly, six bf16
// products per fp32 product, fp32 accumulation, dropped terms <= 2^-24 relative).  Replaces the weight / bias half of
// aten::convolution_backward for nn.Conv2d(k=3, p=1) (reference vgg_osvos.py:41,142; autograd of train_online.py:141).
//
//   D[tap][co][ci] = sum_pixels dY[p][co] * X[p + tap][ci]          (bias gradient = column sums of dY, exact fp32)
//
// The reduction index is the PIXEL axis and a bf16 MFMA wants 8 consecutive k per lane, while the tensors lie [pixel][channel].
// Tiles stay pixel-major in LDS (one 16-byte slot = 8 bf16 channels of one pixel, three planes = the three pieces) and the k-fragments
// are gathered by the LDS itself with ds_read_b64_tr_b16 (semantics verified lane by lane: profiles/r01_tr_b16_probe.txt) -- a tap shift
// is an address offset, nothing is transposed in registers.  Pixel pitch = channel bytes + 64 so the four pixel rows of one gather
// fall on disjoint quarters of the 64 banks.
//   * workgroup = 4 waves = 64 couts x 64 cins x 9 taps; wave (wc, wi) owns a 32 x 32 x 9 block = 9 accumulators (144 registers)
//   * patches of 16 x PH pixels (PH = 6: 480p stage heights 480/240/120/60/30 are multiples of 6); a k-step = one 16-pixel patch row
//   * per (k-step, tap row): 3 tap columns x 3 pieces of X gathered (18 reads) -> 18 MFMAs; fragment sets double buffered
//   * the next patch's fp32 pieces are loaded from INSIDE the k-loop (one item per stage) and split / written after the barrier
//   * deterministic: fixed patch -> split assignment, fp32 slabs [split][tap][co][ci] + the shared slab reduce (wgrad_f32.hip)
#include "common.h"
#include "kernels.h"
#include "h2split.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int PW = 16;

// S16 = 1: the skinny shape of side_prep (Cout = 16; reference vgg_osvos.py:41): the four waves take four 32-cin blocks of ONE 32-cout block
// whose upper 16 couts are zero rows of the dY tile (2x padding instead of the 4x a 64-cout tile would spend): 128 cins x 16 couts per
// workgroup.  The exact-fp32 skinny kernel it replaces runs at 26 TFLOP/s (263 us per step, VERDICT r02 "furthest below any roofline").
// WAVES = 4: 64 couts x 64 cins per workgroup, one wave per SIMD, double-buffered fragment sets.  WAVES = 8 (Cout % 128 == 0): 128 couts x
// 64 cins, TWO waves per SIMD (<= 256 registers each: one fragment set, the partner wave covers the gather latency) -- the X tile and its
// 54 gathers per k-step are shared by twice the MFMAs.
template <int PH_, int WAVES_, int S16_ = 0>
struct G3 {
  static constexpr int PH = PH_, WAVES = WAVES_, NT = 64 * WAVES_, S16 = S16_;
  static constexpr int BCO = S16_ ? 32 : 16 * WAVES_;                  // couts per workgroup (S16: 16 real ones)
  static constexpr int BCI = S16_ ? 128 : 64;                          // cins per workgroup
  static constexpr int XOCT = BCI / 8;                                 // channel octets per X pixel
  static constexpr int DYP = BCO * 2 + 64, XP = BCI * 2 + 64;          // bytes per pixel and plane (+ 64 B skew: conflict-free gathers)
  static constexpr int PPIX = PW * PH, HW_ = PW + 2, XPIX = (PH + 2) * HW_;
  static constexpr int DY_B = PPIX * DYP, X_B = XPIX * XP;             // bytes of one piece plane
  static constexpr int DOCT = BCO / 8;                                 // channel octets per dY pixel
  static constexpr int DY_ITEMS = PPIX * DOCT, X_ITEMS = XPIX * XOCT;  // (pixel, channel octet)
  static constexpr int DPG = NT / DOCT, XPG = NT / XOCT;               // pixels covered per staging round
  static constexpr int NDY = (DY_ITEMS + NT - 1) / NT, NX = (X_ITEMS + NT - 1) / NT, NIT = NDY + NX;
  static constexpr int NST = 3 * PH;                                  // stages (k-step, tap row) per patch
  static constexpr int NB = WAVES_ == 8 ? 1 : 2;                       // fragment sets
  static constexpr size_t LDS = (size_t)3 * (DY_B + X_B);
  static_assert(NIT <= NST, "one staged item per stage must cover the patch");
  static_assert(LDS <= 160 * 1024, "tiles exceed the LDS of a CU");
};

struct W3Args {
  const void* x;       // fp32 NHWC [N][H][W][Cin_s]
  const void* dy;      // fp32 NHWC [N][H][W][Cout_s]
  float* slab;
  float* bslab;
  int N, H, W, Cin_s, Cout, Cout_s;
  int npx, npy, npatches, per_split, nco_t, nci_t;
  int map;
};

__device__ inline unsigned cvt2w(float a, float b) {
  bf16x2_t h;
  h[0] = (__bf16)a;
  h[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, h);
}
__device__ inline void split2w(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
  p0 = cvt2w(a, b);
  float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);
  p1 = cvt2w(ra, rb);
  ra -= __uint_as_float(p1 << 16);
  rb -= __uint_as_float(p1 & 0xffff0000u);
  p2 = cvt2w(ra, rb);
}
__device__ inline void split8w(const u32x4& lo, const u32x4& hi, u32x4& p0, u32x4& p1, u32x4& p2) {
  const f32x4 a = __builtin_bit_cast(f32x4, lo), b = __builtin_bit_cast(f32x4, hi);
  unsigned q0[4], q1[4], q2[4];
  split2w(a[0], a[1], q0[0], q1[0], q2[0]);
  split2w(a[2], a[3], q0[1], q1[1], q2[1]);
  split2w(b[0], b[1], q0[2], q1[2], q2[2]);
  split2w(b[2], b[3], q0[3], q1[3], q2[3]);
  p0 = u32x4{q0[0], q0[1], q0[2], q0[3]};
  p1 = u32x4{q1[0], q1[1], q1[2], q1[3]};
  p2 = u32x4{q2[0], q2[1], q2[2], q2[3]};
}

// scheduling groups of one stage, in program order: MFMA, then what may hide in its 32-cycle shadow -- its share of the NDS gathers and NVM
// staging loads of the stage, a few VALU / SALU (the builtin wants literal sizes: compile-time recursion)
// NM: MFMAs of the stage (18 with three pieces per operand, 9 with two)
template <int NDS, int NVM, int NM = 18, int I = 0>
__device__ __forceinline__ void stage_groups() {
  if constexpr (I < NM) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    constexpr int nds = (NDS * (I + 1)) / NM - (NDS * I) / NM, nvm = (NVM * (I + 1)) / NM - (NVM * I) / NM;
    if constexpr (nds > 0) __builtin_amdgcn_sched_group_barrier(0x100, nds, 0);
    if constexpr (nvm > 0) __builtin_amdgcn_sched_group_barrier(0x020, nvm, 0);
    __builtin_amdgcn_sched_group_barrier(0x006, 3, 0);
    stage_groups<NDS, NVM, NM, I + 1>();
  }
}

// ILV = 1 (four-wave forms): the next stage's 18-24 fragment gathers and the staging loads are INTERLEAVED with the stage's 18 MFMAs (one
// gather per MFMA, sched_group_barrier) instead of being issued as a block in front of them: with one wave per SIMD nothing else covers
// the ~150-250 cycles that block takes while the matrix pipe drains (576 cycles of MFMA per stage; the pipe was busy 51 % of the time).
// NP = 2 (round 6, precision 'fp32x2'): two bf16 pieces per operand, three products (conv3x3_f32x3.hip) -- same tiles, the low planes unused
// HP = 1 ("h2", precision 'fp32h2'; h2split.h): the two pieces are FP16 with a block exponent per operand.  Both operands are activations
// here, so both exponents are the workgroup's own running ones: the waves exchange the largest magnitudes of the patch about to be staged
// (dY and X separately) through two words of LDS before the barrier that opens the staging phase; when an exponent drops the nine accumulators
// are multiplied by the power of two that separates the scales; the slab epilogue un-scales.  (The bias gradient stays the exact fp32 column sum.)
template <int PH, int WAVES, int S16 = 0, int ILV = 0, int NP = 3, int HP = 0>
__global__ __launch_bounds__(64 * WAVES) void wgrad_f32x3_kernel(W3Args a) {
  static_assert(HP == 0 || NP == 2, "h2: two pieces");
  using G = G3<PH, WAVES, S16>;
  constexpr int NT = G::NT, BCO = G::BCO, BCI = G::BCI;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* dYs = smem;                         // [piece 3][PPIX][PITCH]
  char* Xs = smem + 3 * G::DY_B;            // [piece 3][XPIX][PITCH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wc = S16 ? 0 : wave >> 1, wi = S16 ? wave : wave & 1;
  constexpr unsigned OOB = 0x80000000u;

  int id = blockIdx.x;
  if (a.map == 1) id = (id & 7) * (gridDim.x >> 3) + (id >> 3);      // XCD-local: the channel tiles of one split share an L2
  const int cit = id % a.nci_t;
  id /= a.nci_t;
  const int cot = id % a.nco_t;
  const int split = id / a.nco_t;
  const int co0 = cot * BCO, ci0 = cit * BCI;
  const int p_begin = split * a.per_split, p_end = min(p_begin + a.per_split, a.npatches);

  // staged items: (pixel, channel octet) = 32 bytes of fp32 in, 3 x 16 bytes of bf16 pieces out; a thread keeps its octet in every round
  const int doct = tid % G::DOCT, dpg = tid / G::DOCT;      // dY: BCO / 8 octets per pixel
  const int xoct = tid % G::XOCT, xpg = tid / G::XOCT;      // X: BCI / 8 octets per pixel
  const bool dy_ch_ok = co0 + 8 * doct < a.Cout, x_ch_ok = ci0 + 8 * xoct < a.Cin_s;
  u32x4 rdy[G::NDY][2], rx[G::NX][2];
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool want_bias = a.bslab != nullptr && cit == 0;
  const int img_dy_bytes = a.H * a.W * a.Cout_s * 4, img_x_bytes = a.H * a.W * a.Cin_s * 4;

  struct Patch { __amdgpu_buffer_rsrc_t drs, xrs; int x0, y0; };
  auto locate = [&](int p, bool live) -> Patch {
    const int px = p % a.npx;
    int t = p / a.npx;
    const int py = t % a.npy;
    const int n = live ? t / a.npy : 0;
    Patch q;
    q.x0 = live ? px * PW : 0x40000000;            // dead patch: every column test fails -> all loads out of range
    q.y0 = py * PH;
    q.drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(a.dy)) + (size_t)n * img_dy_bytes, 0, img_dy_bytes, 0x00020000);
    q.xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(a.x)) + (size_t)n * img_x_bytes, 0, img_x_bytes, 0x00020000);
    return q;
  };
  // rows above / below the image fall out of the per-image buffer range by themselves (negative offsets wrap past num_records);
  // columns outside the image must be pushed out explicitly (they would alias the neighbouring row)
  auto issue = [&](const Patch& q, int it) {      // it: compile-time item index (dY items first)
    if (it < G::NDY) {
      const int p = dpg + G::DPG * it, py = p / PW, pxx = p - py * PW;
      const bool ok = dy_ch_ok && (G::DY_ITEMS % NT == 0 || p < G::PPIX) && q.x0 + pxx < a.W;
      const unsigned off = ok ? (unsigned)((((q.y0 + py) * a.W + q.x0 + pxx) * a.Cout_s + co0 + 8 * doct) * 4) : OOB;
      rdy[it][0] = __builtin_amdgcn_raw_buffer_load_b128(q.drs, off, 0, 0);
      rdy[it][1] = __builtin_amdgcn_raw_buffer_load_b128(q.drs, off + 16u, 0, 0);
    } else if (it < G::NIT) {
      const int j = it - G::NDY;
      const int hp = xpg + G::XPG * j, hy = hp / G::HW_, hx = hp - hy * G::HW_;
      const bool ok = x_ch_ok && hp < G::XPIX && (unsigned)(q.x0 - 1 + hx) < (unsigned)a.W;
      const unsigned off = ok ? (unsigned)((((q.y0 - 1 + hy) * a.W + q.x0 - 1 + hx) * a.Cin_s + ci0 + 8 * xoct) * 4) : OOB;
      rx[j][0] = __builtin_amdgcn_raw_buffer_load_b128(q.xrs, off, 0, 0);
      rx[j][1] = __builtin_amdgcn_raw_buffer_load_b128(q.xrs, off + 16u, 0, 0);
    }
  };
  int h2_ed = kH2NoScale, h2_ex = kH2NoScale;      // h2: running block exponents of dY and X ...
  float h2_sd = 1.f, h2_sx = 1.f;                  // ... and their powers of two
  unsigned* const h2_mx = reinterpret_cast<unsigned*>(smem + G::LDS - 128);      // [parity 2][operand 2][wave <= 8] words at the end of X's unused third plane
  static_assert(HP == 0 || (G::X_B >= 128 && WAVES <= 8), "h2: exchange slots");
  auto store_patch = [&]() {
#pragma unroll
    for (int i = 0; i < G::NDY; ++i) {
      u32x4 p0, p1, p2;
      if (want_bias) {                            // bias gradient: exact fp32 column sums of dY (zeros outside the image)
        const f32x4 lo = __builtin_bit_cast(f32x4, rdy[i][0]), hi = __builtin_bit_cast(f32x4, rdy[i][1]);
#pragma unroll
        for (int c = 0; c < 4; ++c) { bsum[c] += lo[c]; bsum[4 + c] += hi[c]; }
      }
      if constexpr (HP != 0) h2_split8(rdy[i][0], rdy[i][1], h2_sd, p0, p1);
      else split8w(rdy[i][0], rdy[i][1], p0, p1, p2);
      const int p = dpg + G::DPG * i;
      if (G::DY_ITEMS % NT == 0 || p < G::PPIX) {
        char* d = dYs + p * G::DYP + doct * 16;
        *reinterpret_cast<u32x4*>(d) = p0;
        *reinterpret_cast<u32x4*>(d + G::DY_B) = p1;
        if constexpr (NP == 3) *reinterpret_cast<u32x4*>(d + 2 * G::DY_B) = p2;
      }
    }
#pragma unroll
    for (int j = 0; j < G::NX; ++j) {
      u32x4 p0, p1, p2;
      if constexpr (HP != 0) h2_split8(rx[j][0], rx[j][1], h2_sx, p0, p1);
      else split8w(rx[j][0], rx[j][1], p0, p1, p2);
      const int hp = xpg + G::XPG * j;
      if (G::X_ITEMS % NT == 0 || hp < G::XPIX) {
        char* d = Xs + hp * G::XP + xoct * 16;
        *reinterpret_cast<u32x4*>(d) = p0;
        *reinterpret_cast<u32x4*>(d + G::X_B) = p1;
        if constexpr (NP == 3) *reinterpret_cast<u32x4*>(d + 2 * G::X_B) = p2;
      }
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // fragment gather: lane (fi = lane & 15, fg = (lane >> 4) & 1, lh = lane >> 5) addresses pixel 8 lh + fi / 4 (+4 for the second read),
  // channels 16 fg + 4 (fi % 4) .. +3 of its wave's 32-channel block, and receives channel 16 fg + fi of pixels 8 lh .. 8 lh + 7
  const int fi = lane & 15, fg = (lane >> 4) & 1, lh = lane >> 5;
  const char* a_base = dYs + (8 * lh + (fi >> 2)) * G::DYP + (32 * wc + 16 * fg + 4 * (fi & 3)) * 2;
  const char* b_base = Xs + (8 * lh + (fi >> 2)) * G::XP + (32 * wi + 16 * fg + 4 * (fi & 3)) * 2;
  auto tr8 = [&](const char* p, int pitch) -> s16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * pitch));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };

  {
    const Patch q = locate(p_begin, p_begin < p_end);
#pragma unroll
    for (int it = 0; it < G::NIT; ++it) issue(q, it);
  }
  for (int p = p_begin; p < p_end; ++p) {
    if constexpr (HP != 0) {           // this wave's largest |raw value| of the patch about to be staged, per operand
      unsigned md = 0, mxx = 0;
#pragma unroll
      for (int i = 0; i < G::NDY; ++i) md = h2_amax8(rdy[i][0], rdy[i][1], md);
#pragma unroll
      for (int j = 0; j < G::NX; ++j) mxx = h2_amax8(rx[j][0], rx[j][1], mxx);
      md = h2_wave_max(md);
      mxx = h2_wave_max(mxx);
      if (lane == 0) { h2_mx[(p & 1) * 16 + wave] = md; h2_mx[(p & 1) * 16 + 8 + wave] = mxx; }
    }
    __syncthreads();                   // every wave is done with the previous patch's tiles
    if constexpr (HP != 0) {
      const uint4* q = reinterpret_cast<const uint4*>(h2_mx) + (p & 1) * 4;
      const uint4 d0 = q[0], d1 = q[1], x0 = q[2], x1 = q[3];
      unsigned md = max(max(d0.x, d0.y), max(d0.z, d0.w)), mxx = max(max(x0.x, x0.y), max(x0.z, x0.w));
      if (WAVES == 8) { md = max(md, max(max(d1.x, d1.y), max(d1.z, d1.w))); mxx = max(mxx, max(max(x1.x, x1.y), max(x1.z, x1.w))); }
      const int ed = h2_exp(__builtin_amdgcn_readfirstlane(md)), ex = h2_exp(__builtin_amdgcn_readfirstlane(mxx));
      const int nd = ed < h2_ed ? ed : h2_ed, nx_ = ex < h2_ex ? ex : h2_ex;
      if (nd != h2_ed || nx_ != h2_ex) {          // (uniform) a larger value than any so far: re-express the accumulators at the new scales
        const int dl = (h2_ed == kH2NoScale ? 0 : nd - h2_ed) + (h2_ex == kH2NoScale ? 0 : nx_ - h2_ex);
        if (dl != 0) {
#pragma unroll
          for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = __builtin_ldexpf(acc[t][r], dl);
        }
        h2_ed = nd; h2_ex = nx_;
        h2_sd = h2_pow2(nd); h2_sx = h2_pow2(nx_);
      }
    }
    store_patch();
    __syncthreads();
    const Patch nx = locate(p + 1, p + 1 < p_end);
    constexpr int NB = G::NB;
    s16x8 af[NB][3], bfr[NB][3][3];      // [set][piece] / [set][piece][tap column]
    auto lda = [&](int ks) {
#pragma unroll
      for (int pc = 0; pc < NP; ++pc) af[ks & (NB - 1)][pc] = tr8(a_base + pc * G::DY_B + ks * PW * G::DYP, G::DYP);
    };
    auto ldb = [&](int st) {
      const int ks = st / 3, r = st % 3;
#pragma unroll
      for (int pc = 0; pc < NP; ++pc)
#pragma unroll
        for (int s = 0; s < 3; ++s) bfr[st & (NB - 1)][pc][s] = tr8(b_base + pc * G::X_B + ((ks + r) * G::HW_ + s) * G::XP, G::XP);
    };
    if constexpr (NB == 2) {
      lda(0);
      ldb(0);
    }
#pragma unroll
    for (int st = 0; st < G::NST; ++st) {
      const int ks = st / 3, r = st % 3;
      if constexpr (NB == 2) {
        if (st + 1 < G::NST) {
          if (r == 2) lda(ks + 1);
          ldb(st + 1);
        }
      } else {      // two waves per SIMD cover each other's gather latency; 256 registers leave no room for a second set
        if (r == 0) lda(ks);
        ldb(st);
      }
      issue(nx, st);                    // next patch's fp32 pieces: one item (two 16-byte loads) per stage
      if constexpr (ILV == 0) __builtin_amdgcn_sched_barrier(0);
      // pieces: 0 = high, 1 = middle, 2 = low; small products first.  First operand = X (rows = cin), second = dY (columns = cout)
      constexpr int NPROD = NP == 3 ? 6 : 3;
      constexpr int PX[6] = {NP == 3 ? 2 : 1, 0, NP == 3 ? 1 : 0, 1, 0, 0};      // NP = 2: (xm, dh), (xh, dm), (xh, dh)
      constexpr int PD[6] = {0, NP == 3 ? 2 : 1, NP == 3 ? 1 : 0, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < NPROD; ++t)
#pragma unroll
        for (int s = 0; s < 3; ++s)
          if constexpr (HP != 0)
            acc[r * 3 + s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, bfr[st & (NB - 1)][PX[t]][s]),
                                                                   __builtin_bit_cast(f16x8_t, af[ks & (NB - 1)][PD[t]]), acc[r * 3 + s], 0, 0, 0);
          else
            acc[r * 3 + s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, bfr[st & (NB - 1)][PX[t]][s]),
                                                                    __builtin_bit_cast(bf16x8_t, af[ks & (NB - 1)][PD[t]]), acc[r * 3 + s], 0, 0, 0);
      if constexpr (ILV != 0) {
        // gathers of the NEXT stage (independent registers: the fragment sets are double buffered) and this stage's staging loads
        constexpr int V = 2;
        const bool last = st + 1 >= G::NST, wide = r == 2, vm = st < G::NIT;      // (compile-time after unrolling)
        constexpr int NM = 3 * NPROD, DN = 6 * NP, DW = 8 * NP;      // MFMAs of a stage; gathers of the next stage (narrow / with the dY fragment)
        if (last) { if (vm) stage_groups<0, V, NM>(); else stage_groups<0, 0, NM>(); }
        else if (wide) { if (vm) stage_groups<DW, V, NM>(); else stage_groups<DW, 0, NM>(); }
        else { if (vm) stage_groups<DN, V, NM>(); else stage_groups<DN, 0, NM>(); }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();
  if constexpr (HP != 0) {
    const int dl = (h2_ed == kH2NoScale || h2_ex == kH2NoScale) ? 0 : -(h2_ed + h2_ex);      // (no patch at all: the accumulators are zero)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = __builtin_ldexpf(acc[t][r], dl);
  }

  {   // slab epilogue: D = [cin rows][cout columns]; lane (li, lh) holds cout li and cins 8 q + 4 lh + (0..3) = one 16-byte store
    const int li = lane & 31;
    const size_t slab_elems = (size_t)9 * a.Cout * a.Cin_s;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(a.slab + (size_t)split * slab_elems, 0, (int)(slab_elems * 4), 0x00020000);
    const int co = co0 + wc * 32 + li;
    const int cib = ci0 + wi * 32 + 4 * lh;
    const unsigned row = co < a.Cout ? (unsigned)(co * a.Cin_s) * 4u : OOB;
    const unsigned tap_stride = (unsigned)(a.Cout * a.Cin_s) * 4u;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ci = cib + 8 * q;
        const unsigned off = ci < a.Cin_s ? row + (unsigned)t * tap_stride + (unsigned)ci * 4u : OOB;
        const f32x4 v = {acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), srs, off, 0, 0);
      }
  }
  if (want_bias) {
    float* red = reinterpret_cast<float*>(smem);          // [DPG pixel groups][BCO channels]
#pragma unroll
    for (int c = 0; c < 8; ++c) red[dpg * BCO + doct * 8 + c] = bsum[c];
    __syncthreads();
    if (tid < BCO) {
      float sum = 0.f;
      for (int g = 0; g < G::DPG; ++g) sum += red[g * BCO + tid];
      if (co0 + tid < a.Cout) a.bslab[(size_t)split * a.Cout + co0 + tid] = sum;
    }
  }
}

struct W3Plan {
  int ph, waves, bco, nco_t, nci_t, npx, npy, npatches, nsplit, per_split;
  size_t slab_floats, bslab_floats;
};

// one workgroup per CU (124-135 KB of LDS): aim at one round of ~256 workgroups, each with a long patch range
W3Plan make_plan3(int N, int H, int W, int Cin_s, int Cout) {
  W3Plan p;
  if (Cout == 16) {      // skinny form (S16): 128 cins x 16 couts per workgroup, 16 x 4 pixel patches
    p.waves = 4; p.bco = 32; p.ph = 4;
    p.nco_t = 1;
    p.nci_t = ceil_div(Cin_s, 128);
    p.npx = ceil_div(W, PW);
    p.npy = ceil_div(H, p.ph);
    p.npatches = N * p.npx * p.npy;
    int want = ceil_div(256, p.nci_t);
    const int max_split = p.npatches / 2 > 0 ? p.npatches / 2 : 1;
    if (want > max_split) want = max_split;
    p.per_split = ceil_div(p.npatches, want);
    p.nsplit = ceil_div(p.npatches, p.per_split);
    p.slab_floats = (size_t)p.nsplit * 9 * Cout * Cin_s;
    p.bslab_floats = (size_t)p.nsplit * Cout;
    return p;
  }
  // (An eight-wave 128 x 64-channel tile was measured in round 2: 3-4 % faster standalone on conv2_2 / conv3_2, level or slower elsewhere --
  //  twice the splits = twice the slab traffic -- and the whole step slower beside the data-gradient kernels, 206 vs 210 frames/s: removed.)
  p.waves = 4;
  p.bco = 16 * p.waves;
  p.ph = (ceil_div(H, 6) * 6 <= ceil_div(H, 4) * 4) ? 6 : 4;
  p.nco_t = ceil_div(Cout, p.bco);
  p.nci_t = ceil_div(Cin_s, 64);
  p.npx = ceil_div(W, PW);
  p.npy = ceil_div(H, p.ph);
  p.npatches = N * p.npx * p.npy;
  int want = ceil_div(256, p.nco_t * p.nci_t);      // (round 5 re-check at step level: 192 -0.9 %, 320 -1.5 %, 384 -1.6 %, 512 -4.5 % against 256; profiles/r05_ab_small.txt)
  const int max_split = p.npatches / 2 > 0 ? p.npatches / 2 : 1;
  if (want > max_split) want = max_split;
  if (want > 256) want = 256;
  p.per_split = ceil_div(p.npatches, want);
  p.nsplit = ceil_div(p.npatches, p.per_split);
  p.slab_floats = (size_t)p.nsplit * 9 * Cout * Cin_s;
  p.bslab_floats = (size_t)p.nsplit * Cout;
  return p;
}

template <int PH, int WAVES, int S16 = 0, int ILV = 0, int NP = 3, int HP = 0>
int launch3(const W3Args& a, long blocks, hipStream_t stream) {
  static bool attr_set_dev[OSVOS_MAX_DEVICES] = {};
  bool& attr_set = attr_set_dev[osvos_current_device()];
  if (!attr_set) {
    OSVOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_f32x3_kernel<PH, WAVES, S16, ILV, NP, HP>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)G3<PH, WAVES, S16>::LDS));
    attr_set = true;
  }
  constexpr size_t lds = G3<PH, WAVES, S16>::LDS;
  hipLaunchKernelGGL((wgrad_f32x3_kernel<PH, WAVES, S16, ILV, NP, HP>), dim3((unsigned)blocks), dim3(64 * WAVES), lds, stream, a);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

}  // namespace

int osvos_wgrad_reduce_launch(const float* slab, const float* bslab, float* dw, float* db, int nsplit, int Cout, int Cin,
                              int Cin_s, int accumulate, hipStream_t stream);

// the wide trunk layers (channel counts multiples of 64, no channel padding); conv1_1 and side_prep keep their exact skinny kernels
bool osvos_wgrad_f32x3_applicable(int Cin, int Cin_s, int Cout, int Cout_s) {
  return Cin == Cin_s && Cin_s % 64 == 0 && Cout % 64 == 0 && Cout_s % 4 == 0;
}
// side_prep's shape (Cout = 16): the S16 form
bool osvos_wgrad_f32x3_skinny_applicable(int Cin, int Cin_s, int Cout, int Cout_s) {
  return Cout == 16 && Cout_s == 16 && Cin == Cin_s && Cin_s % 128 == 0;
}

size_t osvos_wgrad_f32x3_ws_bytes(int N, int H, int W, int Cin_s, int Cout) {
  if (Cout == 16 && Cin_s % 128 == 0) {
    const W3Plan p = make_plan3(N, H, W, Cin_s, Cout);
    return align_up((p.slab_floats + p.bslab_floats) * sizeof(float), 256);
  }
  if (Cin_s % 64 != 0 || Cout % 64 != 0) return 0;
  const W3Plan p = make_plan3(N, H, W, Cin_s, Cout);
  return align_up((p.slab_floats + p.bslab_floats) * sizeof(float), 256);
}

namespace {
int wgrad3_run(const void* x, const void* dy, void* ws, float* dw, float* db,
               int N, int H, int W, int Cin, int Cin_s, int Cout, int Cout_s, int accumulate, hipStream_t stream) {
  OSVOS_ARG_CHECK(x && dy && ws && dw, "wgrad f32x3: null pointer");
  OSVOS_ARG_CHECK(N > 0 && H > 0 && W > 0, "wgrad f32x3: bad shape");
  const bool skinny = Cout == 16 && Cout_s == 16 && Cin == Cin_s && Cin_s % 128 == 0;      // side_prep
  OSVOS_ARG_CHECK(skinny || osvos_wgrad_f32x3_applicable(Cin, Cin_s, Cout, Cout_s),
                  "wgrad f32x3: unsupported shape (Cin %d/%d Cout %d/%d)", Cin, Cin_s, Cout, Cout_s);
  OSVOS_ARG_CHECK((long)H * W * Cin_s < (1L << 28) && (long)H * W * Cout_s < (1L << 28), "wgrad f32x3: image too large for 31-bit byte offsets");
  const W3Plan p = make_plan3(N, H, W, Cin_s, Cout);
  W3Args a;
  a.x = x; a.dy = dy;
  a.slab = reinterpret_cast<float*>(ws);
  a.bslab = db ? a.slab + p.slab_floats : nullptr;
  a.N = N; a.H = H; a.W = W; a.Cin_s = Cin_s; a.Cout = Cout; a.Cout_s = Cout_s;
  a.npx = p.npx; a.npy = p.npy; a.npatches = p.npatches; a.per_split = p.per_split; a.nco_t = p.nco_t; a.nci_t = p.nci_t;
  const long blocks = (long)p.nsplit * p.nco_t * p.nci_t;
  OSVOS_ENV_INT(map_env, "OSVOS_WGRAD_MAP", 1);
  a.map = (map_env == 1 && blocks % 8 == 0) ? 1 : 0;
  const int phase = osvos_wgrad_phase();
  if (phase != 2) {
    // gathers / staging loads interleaved with the MFMAs of the previous stage (round 3; the block-issue form measured level and is gone)
    const bool two = osvos_x3_pieces() == 2;      // precision 'fp32x2'
    const bool h2 = osvos_x3_pieces() == 22;      // precision 'fp32h2'
    const int rc = h2  ? (skinny ? launch3<4, 4, 1, 1, 2, 1>(a, blocks, stream)
                                 : (p.ph == 6 ? launch3<6, 4, 0, 1, 2, 1>(a, blocks, stream) : launch3<4, 4, 0, 1, 2, 1>(a, blocks, stream)))
                 : two ? (skinny ? launch3<4, 4, 1, 1, 2>(a, blocks, stream)
                                 : (p.ph == 6 ? launch3<6, 4, 0, 1, 2>(a, blocks, stream) : launch3<4, 4, 0, 1, 2>(a, blocks, stream)))
                       : (skinny ? launch3<4, 4, 1, 1>(a, blocks, stream)
                                 : (p.ph == 6 ? launch3<6, 4, 0, 1>(a, blocks, stream) : launch3<4, 4, 0, 1>(a, blocks, stream)));
    if (rc) return rc;
  }
  if (phase == 1) return 0;
  return osvos_wgrad_reduce_launch(a.slab, a.bslab, dw, db, p.nsplit, Cout, Cin, Cin_s, accumulate, stream);
}
}  // namespace

int osvos_conv3x3_wgrad_f32x3(const float* x, const float* dy, void* ws, float* dw, float* db,
                              int N, int H, int W, int Cin, int Cin_s, int Cout, int Cout_s, int accumulate, hipStream_t stream) {
  return wgrad3_run(x, dy, ws, dw, db, N, H, W, Cin, Cin_s, Cout, Cout_s, accumulate, stream);
}
