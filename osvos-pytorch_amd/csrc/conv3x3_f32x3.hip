// 3x3 stride-1 pad-1 convolution, NHWC fp32 in / fp32 out, computed on the bf16 matrix pipe with THREE-WAY SPLIT
// operands ("f32x3"): the same problem, the same tensors and the same weight packs as conv3x3_f32.hip
// (reference vgg_osvos.py:41,142-143 forward; with the rotated pack, the data-gradient half of its backward).
//
// Why: on gfx950 v_mfma_f32_32x32x2_f32 peaks at 157 TFLOP/s while v_mfma_f32_32x32x16_bf16 peaks at 2.5 PFLOP/s,
// 16x more.  An fp32 number is the exact sum of three bf16 numbers (8 + 8 + 8 significand bits, each piece the
// round-to-nearest-even bf16 of what is left: v = h + m + l), so
//     a * b = ah*bh + ah*bm + am*bh + am*bm + ah*bl + al*bh  (+ am*bl + al*bm + al*bl, each <= 2^-24 |a b|, dropped)
// i.e. SIX bf16 MFMAs with fp32 accumulation reproduce an fp32 product to fp32 round-off (every bf16 x bf16
// product is exact in fp32; the accumulator is the same fp32 register an fp32 MFMA would use).  Six passes over a
// pipe that is 16x faster = 2.67x the fp32-MFMA rate for fp32-grade results.  The split happens while the fp32
// tiles are staged into LDS (v_cvt_pk_bf16_f32 + two subtractions per piece); HBM and L2 see plain fp32 tensors.
// tests/test_gpu_ops.py holds this kernel to the SAME float64 bars as the exact fp32 kernel.
//
// Structure (implicit GEMM, M = pixels, N = Cout, K = 9 * Cin, 16-channel K chunks = one MFMA k-step):
//   * halo tile (TH+2) x (TW+2) staged once per chunk as 3 planes x 2 groups of 16-byte slots (8 bf16 channels of one
//     pixel), re-used by all 9 taps (a tap is an LDS address offset); weights of the chunk (9 taps x 16 ci x BN co)
//     staged from the fp32 pack [tap][Cin/4][CoutP][4] and split the same way
//   * one LDS buffer, the NEXT chunk waits in registers (raw fp32, loaded with branch-free buffer loads while the
//     MFMAs of the current chunk run)
//   * per (tap, M block): 3 A fragments + (per tap) 3 x WN B fragments -> 6 WN MFMAs; fragment reads are issued one
//     step ahead and pinned with sched_barrier (hipcc otherwise sinks every ds_read to its first use)
//   * epilogue identical to conv3x3_f32.hip (cout-major accumulators -> 16-byte buffer stores, bias / ReLU / ReLU-mask
//     of the producer fused, split-K partial sums)
//
// STREAM-K form (SK = 1; round 4).  At batch 1 the 256-pixel tiles give 202-224 workgroups on 256 CUs (conv3_x / conv4_x), 420 on conv2_x:
// one workgroup per CU (110-143 KB of LDS), so 12-18 % of the chip idles for lack of tiles and nothing runs beside the forward.  The SK kernel
// is launched with one PERSISTENT workgroup per CU; the launch's work = ntiles x (Cin / 16) units (tile, 16-channel K chunk) in one linear
// order is cut into gridDim.x equal contiguous shares.  A workgroup walks its share tile by tile; a tile whose K range it covers alone gets the
// normal epilogue, otherwise the raw accumulators go to a partial slot in the workspace ([virtual workgroup][2] slots: a share has at most one
// partial head and one partial tail), the workgroup takes a ticket on the tile, and the LAST arriver sums the parts IN CONTRIBUTOR ORDER (its
// own from registers: the sum does not depend on who arrives last -> deterministic) and runs the epilogue.  No spin-waits: a workgroup never
// waits for another one, so no residency assumption is made.  Cross-CU visibility follows MI355X_MICROARCH.md "Workgroup dispatch ...": plain
// stores -> every wave drains its stores -> barrier -> lane 0: agent release fence + s_waitcnt vmcnt(0) + relaxed agent atomic; the last
// arriver: agent acquire fence by lane 0 -> barrier -> plain loads.  Blocks are renumbered so that XCD b % 8 owns a CONTIGUOUS eighth of the
// unit space: with the spatial tile as the fast index an XCD keeps one slice of the weights hot in its L2 (deep layers), with the Cout tile
// as the fast index the Cout tiles of one halo run back to back in one workgroup (shallow layers).
#include "common.h"
#include "kernels.h"
#include "epi.h"
#include "maskbits.h"
#include "h2split.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct ConvArgsX {
  const float* x;
  const float* wpk;      // fp32 pack [9][Cin/4][CoutP][4] (osvos_pack_fwd_f32 / osvos_pack_dgrad_f32)
  const uint4* wpk3;     // PS kernels: pre-split pack [piece 3][9][Cin/8][CoutP][8 bf16] (osvos_pack_x3)
  const float* bias;
  const float* mask;
  float* y;
  int N, H, W, Cin, Cout, CoutP, y_cs;
  int tiles_x, tiles_y, nct;
  int relu, map, nsp;
  int ksplit;
  float* part;
  ConvEpi epi;           // optional fused pooling epilogues (epi.h)
  // stream-K kernels only
  int sk_order;          // 0: Cout tile is the fast index of the tile order, 1: the spatial tile is
  unsigned* sk_tickets;  // [ntiles], zero at launch; the last arriver of a tile zeroes its ticket again
  float* sk_part;        // [gridDim.x][2] slots of NT * WM * WN * 16 floats (raw accumulators)
};

constexpr int cdivx(int a, int b) { return (a + b - 1) / b; }
constexpr int pitch_x(int rbw, int hw) {
  return rbw == 32 ? hw : (rbw == 16 ? cdivx(hw, 16) * 16 : cdivx(hw - 8, 16) * 16 + 8);
}

template <int RBW_, int TBX_, int TBY_, int NB_, int WGM_, int WGN_, int OCC_, int ILV_ = 0>
struct CfgX {
  static constexpr int RBW = RBW_, TBX = TBX_, TBY = TBY_, NB = NB_, WGM = WGM_, WGN = WGN_, OCC = OCC_;
  static constexpr int ILV = ILV_;        // 1: the next chunk's global loads are issued one item per (tap, M block) step inside the MFMA loop
  static constexpr int RBH = 32 / RBW;
  static constexpr int TW = TBX * RBW, TH = TBY * RBH;
  static constexpr int HWD = TW + 2, HHT = TH + 2;
  static constexpr int PITCH = pitch_x(RBW, HWD);
  static constexpr int PLANE = HHT * PITCH;              // 16-byte slots of one (piece, channel group) plane of the halo tile
  static constexpr int BN = NB * 32;
  static constexpr int A_U4 = 6 * PLANE;                 // [piece 3][group 2][PLANE]
  static constexpr int B_ITEMS = 18 * BN;                // [tap 9][group 2][BN]
  static constexpr int B_U4 = 3 * B_ITEMS;               // [piece 3][tap 9][group 2][BN]
  static constexpr int BUF_U4 = A_U4 + B_U4;
  static constexpr int A_ITEMS = HHT * HWD * 2;
  static constexpr int NT = 64 * WGM * WGN;
  static constexpr int NA = cdivx(A_ITEMS, NT);
  static constexpr int NBL = cdivx(B_ITEMS, NT);
  static constexpr int MB = TBX * TBY;
  static constexpr int WM = MB / WGM, WN = NB / WGN;
  static constexpr size_t LDS_BYTES = (size_t)(BUF_U4 + 2) * 16;      // (+ one slot of slack, + the stream-K ticket word)
  // two pieces per operand (NP = 2): two thirds of the planes (+ four slots for the h2 exchange words) -- the 64-cout tiles then fit a CU TWICE
  static constexpr int A_U4_2 = 4 * PLANE, B_U4_2 = 2 * B_ITEMS;
  static constexpr size_t LDS_BYTES_2 = (size_t)(A_U4_2 + B_U4_2 + 2 + 4) * 16;
  static constexpr int SK_SLOT_F4 = NT * WM * WN * 4;       // float4 per stream-K partial slot
  static_assert(WGM * WGN == 4 || WGM * WGN == 8, "4 or 8 waves per workgroup");
  static_assert(MB % WGM == 0 && NB % WGN == 0, "wave grid must divide the tile");
  static_assert(LDS_BYTES <= 160 * 1024, "tile does not fit the 160 KB LDS of a gfx950 CU");
};

__device__ inline unsigned cvt2(float a, float b) {
  bf16x2_t h;
  h[0] = (__bf16)a;
  h[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, h);
}
// v = p0 + p1 + p2 exactly (pieces are round-to-nearest-even bf16 of the running remainder)
__device__ inline void split2(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
  p0 = cvt2(a, b);
  float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);
  p1 = cvt2(ra, rb);
  ra -= __uint_as_float(p1 << 16);
  rb -= __uint_as_float(p1 & 0xffff0000u);
  p2 = cvt2(ra, rb);
}
__device__ inline void split8(const u32x4& lo, const u32x4& hi, uint4& p0, uint4& p1, uint4& p2) {
  const f32x4 a = __builtin_bit_cast(f32x4, lo), b = __builtin_bit_cast(f32x4, hi);
  split2(a[0], a[1], p0.x, p1.x, p2.x);
  split2(a[2], a[3], p0.y, p1.y, p2.y);
  split2(b[0], b[1], p0.z, p1.z, p2.z);
  split2(b[2], b[3], p0.w, p1.w, p2.w);
}

// the two leading pieces only (NP = 2): v ~ p0 + p1, the remainder (< 2^-16 |v|) dropped
__device__ inline void split8_hm(const u32x4& lo, const u32x4& hi, uint4& p0, uint4& p1) {
  const f32x4 a = __builtin_bit_cast(f32x4, lo), b = __builtin_bit_cast(f32x4, hi);
  auto two = [](float x, float y, unsigned& q0, unsigned& q1) {
    q0 = cvt2(x, y);
    q1 = cvt2(x - __uint_as_float(q0 << 16), y - __uint_as_float(q0 & 0xffff0000u));
  };
  two(a[0], a[1], p0.x, p1.x);
  two(a[2], a[3], p0.y, p1.y);
  two(b[0], b[1], p0.z, p1.z);
  two(b[2], b[3], p0.w, p1.w);
}

// PS = 1: the weights arrive PRE-SPLIT (three bf16 piece planes, made once per optimizer step by pack_x3_kernel): their staging is
// a plain copy.  5 of the 7 items a thread stages per chunk are weights, re-split by every workgroup of every launch when PS = 0 --
// 64 % of the VALU work between the two barriers of a chunk (profiles/r02_pmc_f32x3.txt).
// NP = 3: the three-way split, six products (fp32-grade).  NP = 2 ("f32x2", round 6): only the high and middle pieces are staged and the three
// products ah*bh + ah*bm + am*bh are formed -- operands carry 16 significand bits (bf16 + bf16; TF32, what cuDNN runs fp32 convolutions in by
// default on the reference's own GPUs, carries 11), accumulation stays fp32: half the matrix work, a relative error of ~2^-17 per product instead
// of 2^-24.  Same LDS layout and the same pre-split packs (the low planes are simply not read).  Precision 'fp32x2' of the network; NOT the
// default: it does not hold the flat fp32 parity bars on every head (profiles/r06_fp32x2.txt).
// HP = 1 ("h2", precision 'fp32h2'; h2split.h): the two pieces are FP16 with a block exponent -- 22-23 significand bits per operand instead of 16,
// the same three products on v_mfma_f32_32x32x16_f16.  The weights arrive pre-split AND pre-scaled by the pack (their exponent rides in the
// pack's unused third plane); the activations' exponent is the workgroup's own: before a chunk is split, the waves exchange the largest
// magnitude of the chunk's raw values through LDS (no extra barrier: written before the barrier that opens the staging phase, read after it);
// the running exponent only ever decreases, and when it does the accumulators are multiplied by the (exact) power of two that separates the
// old scale from the new one.  The epilogue un-scales once.
template <class C, int PS, int SK, int NP = 3, int HP = 0>
__global__ __launch_bounds__(C::NT, C::OCC) void conv3x3_f32x3_kernel(ConvArgsX a) {
  static_assert(NP == 2 || NP == 3, "two or three bf16 pieces per operand");
  static_assert(HP == 0 || (NP == 2 && PS == 1 && SK == 0), "h2: two pieces, pre-split pack, plain grid");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* As = reinterpret_cast<uint4*>(smem);
  uint4* Bs = As + (NP == 3 ? C::A_U4 : C::A_U4_2);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave / C::WGN, wn = wave % C::WGN;
  const int nch_all = a.Cin >> 4;

  // stream-K: this workgroup's share [u, u_end) of the U = ntiles * nch_all units; vme = its index in unit order (XCD-contiguous)
  unsigned u = 0, u_end = 0, vme = 0, sk_U = 0, my_first_tile = 0;
  if constexpr (SK != 0) {
    const unsigned G = gridDim.x;
    vme = (G & 7u) == 0 ? (blockIdx.x & 7u) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    sk_U = (unsigned)(a.nct * a.nsp) * (unsigned)nch_all;
    u = (unsigned)(((unsigned long long)vme * sk_U) / G);
    u_end = (unsigned)(((unsigned long long)(vme + 1) * sk_U) / G);
    my_first_tile = u / (unsigned)nch_all;
  }
 for (;;) {      // SK: one pass per (tile, K range) segment of the share; otherwise exactly one pass
  int sp, ct;
  int kc_begin, kc_end;
  unsigned tile_id = 0;
  if constexpr (SK != 0) {
    if (u >= u_end) break;
    tile_id = u / (unsigned)nch_all;
    kc_begin = (int)(u - tile_id * (unsigned)nch_all);
    kc_end = min(nch_all, kc_begin + (int)(u_end - u));
    if (a.sk_order == 0) { sp = (int)(tile_id / (unsigned)a.nct); ct = (int)(tile_id % (unsigned)a.nct); }
    else { ct = (int)(tile_id / (unsigned)a.nsp); sp = (int)(tile_id % (unsigned)a.nsp); }
  } else {
    if (a.map == 0) {          // Cout tile in the low bits: XCD b % 8 keeps one weight slice hot in its L2
      sp = blockIdx.x / a.nct;
      ct = blockIdx.x % a.nct;
    } else {                   // Cout tiles of one spatial tile on the same XCD: the halo is fetched from HBM once per XCD
      const int j = blockIdx.x >> 3;
      ct = j % a.nct;
      sp = (j / a.nct) * 8 + (blockIdx.x & 7);
      if (sp >= a.nsp) return;
    }
    kc_begin = (int)((long)nch_all * blockIdx.y / a.ksplit);
    kc_end = (int)((long)nch_all * (blockIdx.y + 1) / a.ksplit);
  }
  const int tx = sp % a.tiles_x;
  sp /= a.tiles_x;
  const int ty = sp % a.tiles_y;
  const int n = sp / a.tiles_y;
  const int x0 = tx * C::TW, y0 = ty * C::TH, co0 = ct * C::BN;
  const int CQ = a.Cin >> 2;
  const float* ximg = a.x + (size_t)n * a.H * a.W * a.Cin;

  constexpr unsigned OOB = 0x80000000u;      // past num_records: the buffer load returns zeros, no branch in the K loop
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ximg), 0, (int)((size_t)a.H * a.W * a.Cin * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wpk), 0, (int)((size_t)9 * CQ * a.CoutP * 16), 0x00020000);
  unsigned a_off[C::NA];
  int a_dst[C::NA];
#pragma unroll
  for (int i = 0; i < C::NA; ++i) {
    const int e = tid + i * C::NT;
    const int g = e & 1, pix = e >> 1;
    const int hy = pix / C::HWD, hx = pix % C::HWD;
    const int gy = y0 + hy - 1, gx = x0 + hx - 1;
    const bool slot = e < C::A_ITEMS;
    a_dst[i] = slot ? g * C::PLANE + hy * C::PITCH + hx : -1;
    a_off[i] = (slot && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? (unsigned)(((gy * a.W + gx) * a.Cin + 8 * g) * 4) : OOB;
  }
  constexpr int NBI = PS ? cdivx(NP * C::B_ITEMS, C::NT) : C::NBL;      // weight items per thread and chunk
  constexpr int TPI = C::NT / (2 * C::BN);                              // PS: (piece, tap) rows a thread advances per item
  static_assert(!PS || C::NT % (2 * C::BN) == 0, "pre-split staging: the thread block must cover whole (piece, tap) rows");
  unsigned b_off[PS ? 1 : C::NBL];
  unsigned b_step = 0;                                                  // PS: byte distance between a thread's consecutive items (uniform)
  int b_row0 = 0;
  if (PS) {
    const int CG = a.Cin >> 3;
    b_row0 = tid / (2 * C::BN);
    const int rem = tid % (2 * C::BN), g = rem / C::BN, nn = rem % C::BN;
    b_off[0] = (co0 + nn < a.CoutP) ? (unsigned)(((b_row0 * CG + g) * a.CoutP + co0 + nn) * 16) : OOB;
    b_step = (unsigned)(TPI * CG * a.CoutP * 16);
  } else {
#pragma unroll
    for (int i = 0; i < C::NBL; ++i) {
      const int e = tid + i * C::NT;
      const int tap = e / (2 * C::BN), rem = e % (2 * C::BN);
      const int g = rem / C::BN, nn = rem % C::BN;
      b_off[i] = (e < C::B_ITEMS && co0 + nn < a.CoutP) ? (unsigned)(((tap * CQ + 2 * g) * a.CoutP + co0 + nn) * 16) : OOB;
    }
  }
  const unsigned b_q1 = (unsigned)a.CoutP * 16u;       // the second channel quad of a group sits one [CoutP][4] row further
  const __amdgpu_buffer_rsrc_t w3rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(PS ? a.wpk3 : reinterpret_cast<const uint4*>(a.wpk)), 0,
                                                                        PS ? (int)((size_t)27 * (a.Cin >> 3) * a.CoutP * 16) : 0, 0x00020000);

  u32x4 ra[C::NA][2], rb[NBI][PS ? 1 : 2];
  auto load_item = [&](int it, int kc) {      // it: compile-time item index (A items first)
    if (it < C::NA) {
      ra[it][0] = __builtin_amdgcn_raw_buffer_load_b128(xrs, a_off[it], kc * 64, 0);
      ra[it][1] = __builtin_amdgcn_raw_buffer_load_b128(xrs, a_off[it] + 16u, kc * 64, 0);
    } else if (it < C::NA + NBI) {
      const int i = it - C::NA;
      if constexpr (PS != 0) {
        const unsigned off = (b_row0 + i * TPI < 9 * NP) ? b_off[0] : OOB;   // (only a thread's last item can fall off the 9 NP (piece, tap) rows)
        rb[i][0] = __builtin_amdgcn_raw_buffer_load_b128(w3rs, off, kc * 2 * a.CoutP * 16 + i * b_step, 0);
      } else {
        rb[i][0] = __builtin_amdgcn_raw_buffer_load_b128(wrs, b_off[i], kc * 4 * a.CoutP * 16, 0);
        rb[i][1] = __builtin_amdgcn_raw_buffer_load_b128(wrs, b_off[i] + b_q1, kc * 4 * a.CoutP * 16, 0);
      }
    }
  };
  auto load_chunk = [&](int kc) {
#pragma unroll
    for (int it = 0; it < C::NA + NBI; ++it) load_item(it, kc);
  };
  static_assert(!C::ILV || C::NA + NBI <= 9 * C::WM, "interleaved staging: one item per step must cover the chunk");
  int h2_ea = kH2NoScale;                 // h2: the tile's running block exponent ...
  float h2_sc = 1.f;                      // ... and 2^h2_ea
  unsigned* const h2_mx = reinterpret_cast<unsigned*>(As + C::A_U4_2 + C::B_U4_2 + 2);      // [parity 2][wave 8] words behind the two-piece tiles
  static_assert(HP == 0 || C::NT <= 512, "h2: exchange slots");
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < C::NA; ++i) {
      uint4 p0, p1, p2;
      if constexpr (HP != 0) h2_split8(ra[i][0], ra[i][1], h2_sc, p0, p1);
      else if constexpr (NP == 3) split8(ra[i][0], ra[i][1], p0, p1, p2);
      else split8_hm(ra[i][0], ra[i][1], p0, p1);
      if (C::A_ITEMS % C::NT == 0 || a_dst[i] >= 0) {
        As[a_dst[i]] = p0;
        As[a_dst[i] + 2 * C::PLANE] = p1;
        if constexpr (NP == 3) As[a_dst[i] + 4 * C::PLANE] = p2;
      }
    }
    if constexpr (PS != 0) {
#pragma unroll
      for (int i = 0; i < NBI; ++i) {
        const int e3 = tid + i * C::NT;                 // LDS image [piece][tap][group][BN] = the pack's order
        if ((NP * C::B_ITEMS) % C::NT == 0 || e3 < NP * C::B_ITEMS) Bs[e3] = __builtin_bit_cast(uint4, rb[i][0]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NBI; ++i) {
        uint4 p0, p1, p2;
        if constexpr (NP == 3) split8(rb[i][0], rb[i][1], p0, p1, p2);
        else split8_hm(rb[i][0], rb[i][1], p0, p1);
        const int e = tid + i * C::NT;
        if (C::B_ITEMS % C::NT == 0 || e < C::B_ITEMS) {
          Bs[e] = p0;
          Bs[e + C::B_ITEMS] = p1;
          if constexpr (NP == 3) Bs[e + 2 * C::B_ITEMS] = p2;
        }
      }
    }
  };

  int a_idx[C::WM];
#pragma unroll
  for (int mi = 0; mi < C::WM; ++mi) {
    const int mb = wm * C::WM + mi;
    const int mbx = mb % C::TBX, mby = mb / C::TBX;
    const int dy = li / C::RBW, dx = li % C::RBW;
    a_idx[mi] = lh * C::PLANE + (mby * C::RBH + dy) * C::PITCH + mbx * C::RBW + dx;
  }
  const int b_idx = lh * C::BN + wn * C::WN * 32 + li;

  f32x16 acc[C::WM][C::WN];
#pragma unroll
  for (int mi = 0; mi < C::WM; ++mi)
#pragma unroll
    for (int ni = 0; ni < C::WN; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // probe builds (tools/native/build.sh, -DOSVOS_X3_ABL=n; wrong results, timing only): 1 no MFMA, 2 no fragment reads after the first step,
  // 3 no global loads inside the K loop, 4 tiles stored once (no split / ds_write per chunk), 5 no barriers
  load_chunk(kc_begin);
  // (s_setprio was measured in round 5 -- static priority 1 for the second wave of every SIMD, and priority 1 during a chunk's MFMAs / 0 during
  //  its split + store phase: both within +-0.3 % of no priority at step level, profiles/r05_ab_setprio.txt -- and is not in the kernel)
  for (int kc = kc_begin; kc < kc_end; ++kc) {
    if constexpr (HP != 0) {             // this wave's largest |raw value| of the chunk about to be staged
      unsigned m = 0;
#pragma unroll
      for (int i = 0; i < C::NA; ++i) m = h2_amax8(ra[i][0], ra[i][1], m);
      m = h2_wave_max(m);
      if (lane == 0) h2_mx[(kc & 1) * 8 + wave] = m;
    }
#if !(defined(OSVOS_X3_ABL) && OSVOS_X3_ABL == 5)
    __syncthreads();                     // every wave is done with the previous chunk's tiles
#endif
    if constexpr (HP != 0) {
      const uint4 m0 = reinterpret_cast<const uint4*>(h2_mx)[(kc & 1) * 2], m1 = reinterpret_cast<const uint4*>(h2_mx)[(kc & 1) * 2 + 1];
      unsigned m = max(max(max(m0.x, m0.y), max(m0.z, m0.w)), max(max(m1.x, m1.y), max(m1.z, m1.w)));
      if (C::NT < 512) m = max(max(m0.x, m0.y), max(m0.z, m0.w));
      const int e_new = h2_exp(__builtin_amdgcn_readfirstlane(m));
      if (e_new < h2_ea) {               // (uniform) a larger value than any so far: re-express the accumulators at the new scale
        if (h2_ea != kH2NoScale) {
          const int dl = e_new - h2_ea;
#pragma unroll
          for (int mi = 0; mi < C::WM; ++mi)
#pragma unroll
            for (int ni = 0; ni < C::WN; ++ni)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[mi][ni][r] = __builtin_ldexpf(acc[mi][ni][r], dl);
        }
        h2_ea = e_new;
        h2_sc = h2_pow2(e_new);
      }
    }
#if defined(OSVOS_X3_ABL) && OSVOS_X3_ABL == 4
    if (kc == kc_begin)
#endif
    store_chunk();
#if !(defined(OSVOS_X3_ABL) && OSVOS_X3_ABL == 5)
    __syncthreads();
#endif
    const bool more = kc + 1 < kc_end;
    if (!C::ILV && more) load_chunk(kc + 1);      // in flight during the MFMAs below
    // 9 x WM steps (tap, M block); fragments of step s+1 (and, once per tap, the weights of tap+1) are requested
    // before the 6 WN MFMAs of step s issue
    uint4 fb[2][3][C::WN], fa[2][3];
    auto ldB = [&](int tap, int set) {
#if defined(OSVOS_X3_ABL) && OSVOS_X3_ABL == 2
      if (tap > 0) return;
#endif
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int ni = 0; ni < C::WN; ++ni) fb[set][p][ni] = Bs[b_idx + (p * 9 + tap) * 2 * C::BN + ni * 32];
    };
    auto ldA = [&](int tap, int mi, int set) {
#if defined(OSVOS_X3_ABL) && OSVOS_X3_ABL == 2
      if (tap > 0 || mi > 1) return;
#endif
      const int r = tap / 3, s = tap % 3;
#pragma unroll
      for (int p = 0; p < NP; ++p) fa[set][p] = As[a_idx[mi] + p * 2 * C::PLANE + r * C::PITCH + s];
    };
    ldB(0, 0);
    ldA(0, 0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
      for (int mi = 0; mi < C::WM; ++mi) {
        const int step = tap * C::WM + mi;
        if (mi + 1 < C::WM) ldA(tap, mi + 1, (step + 1) & 1);
        else if (tap + 1 < 9) ldA(tap + 1, 0, (step + 1) & 1);
        if (mi == 0 && tap + 1 < 9) ldB(tap + 1, (tap + 1) & 1);
#if !(defined(OSVOS_X3_ABL) && OSVOS_X3_ABL == 3)
        if (C::ILV && more) load_item(step, kc + 1);
#endif
        __builtin_amdgcn_sched_barrier(0);
        const int sa = step & 1, sb = tap & 1;
        // pieces: 0 = high, 1 = middle, 2 = low.  Small products first, the dominant hi x hi product last.
        constexpr int NPROD = NP == 3 ? 6 : 3;
        constexpr int PB[6] = {NP == 3 ? 2 : 1, NP == 3 ? 0 : 0, NP == 3 ? 1 : 0, 1, 0, 0};      // NP = 2: (bm, ah), (bh, am), (bh, ah)
        constexpr int PA[6] = {NP == 3 ? 0 : 0, NP == 3 ? 2 : 1, NP == 3 ? 1 : 0, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < NPROD; ++t)
#pragma unroll
          for (int ni = 0; ni < C::WN; ++ni)
#if defined(OSVOS_X3_ABL) && OSVOS_X3_ABL == 1
            acc[mi][ni][t] += __uint_as_float(fb[sb][PB[t]][ni].x ^ fb[sb][PB[t]][ni].w ^ fa[sa][PA[t]].x ^ fa[sa][PA[t]].w);
#else
            if constexpr (HP != 0)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, fb[sb][PB[t]][ni]),
                                                                   __builtin_bit_cast(f16x8_t, fa[sa][PA[t]]), acc[mi][ni], 0, 0, 0);
            else
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb[sb][PB[t]][ni]),
                                                                    __builtin_bit_cast(bf16x8_t, fa[sa][PA[t]]), acc[mi][ni], 0, 0, 0);
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  if constexpr (HP != 0) {      // un-scale: the pack's exponent (word 0 of its third plane) + the tile's
    const unsigned wbits = reinterpret_cast<const unsigned*>(a.wpk3 + (size_t)2 * 9 * (a.Cin >> 3) * a.CoutP)[0];
    const int dl = -(h2_ea + h2_exp(wbits));
#pragma unroll
    for (int mi = 0; mi < C::WM; ++mi)
#pragma unroll
      for (int ni = 0; ni < C::WN; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = __builtin_ldexpf(acc[mi][ni][r], dl);
  }
  if constexpr (SK != 0) {
    u += (unsigned)(kc_end - kc_begin);
    if (!(kc_begin == 0 && kc_end == nch_all)) {      // this tile's K range is shared with other workgroups
      const unsigned G = gridDim.x, nch = (unsigned)nch_all;
      const unsigned vf = ((tile_id * nch + 1u) * G - 1u) / sk_U;        // owner of the tile's first unit ...
      const unsigned vl = ((tile_id + 1u) * nch * G - 1u) / sk_U;        // ... and of its last one: contributors vf .. vl, in K order
      const unsigned nparts = vl - vf + 1u, cme = vme - vf;
      // Partial slots are written with sc1 (write-through) stores and read with sc1 loads: agent-scope coherent without the L2-wide
      // write-back / invalidate of a release / acquire fence pair (MI355X_MICROARCH.md: "16-B sc1 stores + drained flag", publish-large:
      // 3.0 vs 8.2 us per 64 KB -- and the first form of this kernel, with plain stores + fences, LOST 11 % to the plain grid on conv3_2:
      // every fence flushed an L2 full of other workgroups' freshly written output tiles).
      constexpr int SC1 = 16;
      const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(a.sk_part, 0, (int)((size_t)G * 2 * C::SK_SLOT_F4 * 16), 0x00020000);
      unsigned* const flag = reinterpret_cast<unsigned*>(As + C::BUF_U4 + 1);
      // If every other contributor has already arrived, this workgroup is the last one whatever it does: it keeps its part in registers and
      // skips the publish (the usual case for the workgroup that holds a tile's LAST K range as the first segment of its share).
      if (tid == 0) *flag = __hip_atomic_load(a.sk_tickets + tile_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      bool last = *flag == nparts - 1u;
      if (!last) {
        const unsigned mine = (2u * vme + (tile_id == my_first_tile ? 0u : 1u)) * (unsigned)(C::SK_SLOT_F4 * 16);
#pragma unroll
        for (int mi = 0; mi < C::WM; ++mi)
#pragma unroll
          for (int ni = 0; ni < C::WN; ++ni)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x4 v = {acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]};
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), prs, mine + (unsigned)((((mi * C::WN + ni) * 4 + q) * C::NT + tid) * 16), 0, SC1);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's part has been written through
        __syncthreads();
        if (tid == 0) *flag = __hip_atomic_fetch_add(a.sk_tickets + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        last = *flag == nparts - 1u;
        if (!last) continue;                                  // on to the next segment (uniform)
      }
      if (tid == 0) a.sk_tickets[tile_id] = 0u;               // ready for the next launch that uses the workspace
      // sum of the parts in contributor (= K) order, this workgroup's own part taken from its registers: the same bits whoever arrives last
      f32x16 own[C::WM][C::WN];
#pragma unroll
      for (int mi = 0; mi < C::WM; ++mi)
#pragma unroll
        for (int ni = 0; ni < C::WN; ++ni) own[mi][ni] = acc[mi][ni];
      for (unsigned j = 0; j < nparts; ++j) {
        const bool is_me = j == cme;
        const unsigned vo = vf + j;
        const unsigned first_tile_o = (unsigned)(((unsigned long long)vo * sk_U) / G) / nch;
        const unsigned src = (2u * vo + (tile_id == first_tile_o ? 0u : 1u)) * (unsigned)(C::SK_SLOT_F4 * 16);
        f32x4 part[C::WM][C::WN][4];
#pragma unroll
        for (int mi = 0; mi < C::WM; ++mi)
#pragma unroll
          for (int ni = 0; ni < C::WN; ++ni)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              if (is_me) part[mi][ni][q] = f32x4{own[mi][ni][4 * q], own[mi][ni][4 * q + 1], own[mi][ni][4 * q + 2], own[mi][ni][4 * q + 3]};
              else part[mi][ni][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, src + (unsigned)((((mi * C::WN + ni) * 4 + q) * C::NT + tid) * 16), 0, SC1));
            }
#pragma unroll
        for (int mi = 0; mi < C::WM; ++mi)
#pragma unroll
          for (int ni = 0; ni < C::WN; ++ni)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[mi][ni][4 * q + e] = j == 0 ? part[mi][ni][q][e] : acc[mi][ni][4 * q + e] + part[mi][ni][q][e];
      }
    }
  }
  // ---- epilogue (same as conv3x3_f32.hip): D = [cout rows][pixel columns], lane (li, lh) holds pixel li of its M block
  // and couts 8 q + 4 lh + (0..3) in registers 4q..4q+3 = one 16-byte store
  const size_t img_elems = (size_t)a.H * a.W * a.y_cs;
  const bool split = a.ksplit > 1;       // split-K: raw partial sums, dense [part][n][pixel][Cout]; epilogue in the finalize kernel
  const int cs = split ? a.Cout : a.y_cs;
  const size_t out_elems = (size_t)a.H * a.W * cs;
  float* obase = split ? a.part + ((size_t)blockIdx.y * a.N + n) * out_elems : a.y + n * out_elems;
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(obase, 0, (int)(out_elems * 4), 0x00020000);
  const bool use_mask = !split && a.mask != nullptr;
  const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(use_mask ? a.mask + n * img_elems : a.y), 0,
                                                                       use_mask ? (int)(img_elems * 4) : 0, 0x00020000);
  const bool use_mbits = !split && a.epi.mask_bits != nullptr, put_bits = !split && a.epi.y_bits != nullptr;
  const int bw = a.y_cs >> 5;
  const size_t img_words = (size_t)a.H * a.W * bw;
  const __amdgpu_buffer_rsrc_t mbrs = __builtin_amdgcn_make_buffer_rsrc(use_mbits ? (void*)const_cast<unsigned*>(a.epi.mask_bits + n * img_words) : (void*)a.y, 0,
                                                                        use_mbits ? (int)(img_words * 4) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t ybrs = __builtin_amdgcn_make_buffer_rsrc(put_bits ? (void*)(a.epi.y_bits + n * img_words) : (void*)a.y, 0,
                                                                        put_bits ? (int)(img_words * 4) : 0, 0x00020000);
  const bool use_bias = !split && a.bias != nullptr;
  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(use_bias ? a.bias : a.y), 0, use_bias ? a.Cout * 4 : 0, 0x00020000);
  const bool relu = !split && a.relu;
  float* const anyf = const_cast<float*>(a.wpk != nullptr ? a.wpk : reinterpret_cast<const float*>(a.wpk3));
  // fused pool forward: this launch is the last convolution of a stage (post-ReLU values >= 0, so positions outside the image count as 0)
  const bool pool_fwd = !split && a.epi.pooled != nullptr;
  const int PHo = (a.H + 1) / 2, PWo = (a.W + 1) / 2;
  const size_t poimg = (size_t)PHo * PWo * a.y_cs;
  const __amdgpu_buffer_rsrc_t pors = __builtin_amdgcn_make_buffer_rsrc(pool_fwd ? a.epi.pooled + n * poimg : anyf, 0, pool_fwd ? (int)(poimg * 4) : 0, 0x00020000);
#pragma unroll
  for (int ni = 0; ni < C::WN; ++ni) {
    const int cb = co0 + (wn * C::WN + ni) * 32 + 4 * lh;
    f32x4 bv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bv[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, (cb + 8 * q) * 4, 0, 0));
    f32x4 keep[C::WM][4];      // (pool forward only: the wave's values, zero outside the image)
#pragma unroll
    for (int mi = 0; mi < C::WM; ++mi) {
      const int mb = wm * C::WM + mi;
      const int oy = y0 + (mb / C::TBX) * C::RBH + li / C::RBW;
      const int ox = x0 + (mb % C::TBX) * C::RBW + li % C::RBW;
      const bool inside = oy < a.H && ox < a.W;
      const unsigned pix = inside ? (unsigned)((oy * a.W + ox) * cs) * 4u : OOB;
      const unsigned bitoff = (inside && cb - 4 * lh < a.Cout) ? (unsigned)(oy * a.W + ox) * (unsigned)(bw * 4) + (unsigned)((cb - 4 * lh) >> 5) * 4u : OOB;
      const unsigned mword = mb_load(mbrs, bitoff);
      unsigned ybits = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = cb + 8 * q;
        const unsigned off = co < a.Cout ? pix + (unsigned)co * 4u : OOB;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[mi][ni][4 * q + e] + bv[q][e];
          if (relu) v[e] = v[e] > 0.f ? v[e] : 0.f;
        }
        if (use_mbits) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = mb_test(mword, q, lh, e) ? v[e] : 0.f;
        } else if (use_mask) {
          const f32x4 m = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(mrs, off, 0, 0));
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = m[e] > 0.f ? v[e] : 0.f;
        }
        ybits |= mb_bits_f32(v, q);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, off, 0, 0);
        if (pool_fwd) {
#pragma unroll
          for (int e = 0; e < 4; ++e) keep[mi][q][e] = inside ? v[e] : 0.f;
        }
      }
      if (put_bits) mb_store(ybrs, bitoff, ybits, lh);
    }
    if (pool_fwd) {
      // windows: RBW 32 -- M blocks 2j, 2j+1 of this wave are the two rows, lane ^ 1 the neighbouring column;
      //          RBW 16 -- an M block holds both rows (lanes li and li ^ 16), lane ^ 1 the neighbouring column
      constexpr int NPW = C::RBW == 32 ? C::WM / 2 : C::WM;
#pragma unroll
      for (int j = 0; j < NPW; ++j) {
        const int mb = wm * C::WM + (C::RBW == 32 ? 2 * j : j);
        const int oy = y0 + (mb / C::TBX) * C::RBH, ox = x0 + (mb % C::TBX) * C::RBW + li % C::RBW;      // top row of the window pair
        const bool writer = (li & 1) == 0 && (C::RBW == 32 || li < 16) && oy < a.H && ox < a.W;
        const unsigned ppix = writer ? (unsigned)(((oy >> 1) * PWo + (ox >> 1)) * a.y_cs) * 4u : OOB;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int co = cb + 8 * q;
          f32x4 m;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float t;
            if constexpr (C::RBW == 32) {
              t = fmaxf(keep[2 * j][q][e], keep[2 * j + 1][q][e]);
            } else {
              t = keep[j][q][e];
              t = fmaxf(t, __shfl_xor(t, 16, 64));
            }
            m[e] = fmaxf(t, __shfl_xor(t, 1, 64));
          }
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, m), pors, (co < a.Cout && ppix != OOB) ? ppix + (unsigned)co * 4u : OOB, 0, 0);
        }
      }
    }
  }
  if constexpr (SK == 0) break;
 }
}

constexpr int kSkMaxGrid = 256, kSkMaxTiles = 8192;       // stream-K: workgroups (= CUs of an MI355X) and tickets the workspace is sized for
constexpr size_t kSkSlotBytes = 128 * 1024;               // the largest tile's accumulators (256 pixels x 128 couts fp32)
constexpr size_t kSkTicketBytes = (size_t)kSkMaxTiles * 4;

// sk_grid > 0: the stream-K kernel with that many persistent workgroups (the caller has checked that the tile order has >= sk_grid units)
template <class C, int PS, int SK, int NP = 3, int HP = 0>
int launch_x2(const ConvArgsX& a0, int sk_grid, hipStream_t stream) {
  static bool attr_set_dev[OSVOS_MAX_DEVICES] = {};      // hipFuncSetAttribute is per device
  bool& attr_set = attr_set_dev[osvos_current_device()];
  constexpr size_t lds_bytes = NP == 3 ? C::LDS_BYTES : C::LDS_BYTES_2;
  if (!attr_set) {
    OSVOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_f32x3_kernel<C, PS, SK, NP, HP>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_set = true;
  }
  ConvArgsX a = a0;
  a.tiles_x = ceil_div(a.W, C::TW);
  a.tiles_y = ceil_div(a.H, C::TH);
  a.nct = ceil_div(a.CoutP, C::BN);
  a.nsp = a.tiles_x * a.tiles_y * a.N;
  if constexpr (SK != 0) {
    static_assert((size_t)C::SK_SLOT_F4 * 16 <= kSkSlotBytes, "stream-K partial slot larger than the workspace's");
    const long ntiles = (long)a.nct * a.nsp, units = ntiles * (a.Cin >> 4);
    OSVOS_ARG_CHECK(sk_grid > 0 && sk_grid <= kSkMaxGrid && ntiles <= kSkMaxTiles && units >= sk_grid && a.sk_tickets && a.sk_part && a.ksplit == 1,
                    "conv3x3 f32x3 stream-K: %ld tiles, %ld units on %d workgroups", ntiles, units, sk_grid);
    hipLaunchKernelGGL((conv3x3_f32x3_kernel<C, PS, SK, NP, HP>), dim3((unsigned)sk_grid), dim3(C::NT), lds_bytes, stream, a);
  } else {
    const long blocks = a.map == 0 ? (long)a.nct * a.nsp : (long)a.nct * ((a.nsp + 7) / 8) * 8;
    OSVOS_ARG_CHECK(blocks > 0 && blocks < (1L << 31), "conv3x3 f32x3: grid of %ld blocks", blocks);
    hipLaunchKernelGGL((conv3x3_f32x3_kernel<C, PS, SK, NP, HP>), dim3((unsigned)blocks, (unsigned)a.ksplit), dim3(C::NT), lds_bytes, stream, a);
  }
  OSVOS_LAUNCH_CHECK();
  return 0;
}
// the pre-split form is built for the production tiles (eight waves, in-loop staging); the others take the fp32 pack.  Stream-K: the
// pre-split production tiles of the wide layers (SKT)
template <class C, bool SKT = false>
int launch_x(const ConvArgsX& a, int sk_grid, hipStream_t stream) {
  if (osvos_x3_pieces() == 22) {     // two fp16 pieces with block exponents (precision 'fp32h2'): the pre-split production tiles only
    if constexpr (C::ILV != 0 && (C::NT == 512 || C::NT == 256)) {
      if (a.wpk3 != nullptr) return launch_x2<C, 1, 0, 2, 1>(a, 0, stream);
    }
    osvos_set_error("conv3x3 f32x3: the fp16-pair form is built for the pre-split eight-wave tiles (10, 12, 14, 15, 16, 17) with a pre-split pack");
    return -1;
  }
  if (osvos_x3_pieces() == 2) {      // two-piece mode (precision 'fp32x2'): the plain grid only, no stream-K form
    if constexpr (C::ILV != 0 && (C::NT == 512 || C::NT == 256)) {
      if (a.wpk3 != nullptr) return launch_x2<C, 1, 0, 2>(a, 0, stream);
    }
    OSVOS_ARG_CHECK(a.wpk != nullptr, "conv3x3 f32x3: this tile config has no pre-split form and no fp32 pack was given");
    return launch_x2<C, 0, 0, 2>(a, 0, stream);
  }
  if constexpr (C::ILV != 0 && C::NT == 512) {
    if constexpr (SKT) {
      if (a.wpk3 != nullptr && sk_grid > 0) return launch_x2<C, 1, 1>(a, sk_grid, stream);
    }
    if (a.wpk3 != nullptr) return launch_x2<C, 1, 0>(a, 0, stream);
  }
  OSVOS_ARG_CHECK(a.wpk != nullptr, "conv3x3 f32x3: this tile config has no pre-split form and no fp32 pack was given");
  return launch_x2<C, 0, 0>(a, 0, stream);
}

struct TileInfoX { int tw, th, bn, nt; size_t lds; };

//                 RBW TBX TBY NB WGM WGN OCC
using X0 = CfgX<32, 1, 8, 4, 2, 2, 1>;   // 32x8 px x 128 co, 4 waves (4x2 accumulators per wave), 143 KB LDS
using X1 = CfgX<32, 1, 8, 4, 4, 2, 1>;   // 32x8 px x 128 co, 8 waves (2x2 accumulators per wave): two waves per SIMD
using X2 = CfgX<32, 1, 4, 4, 2, 2, 1>;   // 32x4 px x 128 co, 4 waves (2x2), 130 KB
using X3 = CfgX<32, 1, 8, 2, 2, 2, 1>;   // 32x8 px x  64 co, 4 waves (4x1),  88 KB
using X4 = CfgX<32, 1, 8, 2, 4, 2, 1>;   // 32x8 px x  64 co, 8 waves (2x1)
using X5 = CfgX<32, 1, 4, 2, 2, 2, 2>;   // 32x4 px x  64 co, 4 waves (2x1),  75 KB: two workgroups per CU
using X6 = CfgX<16, 1, 4, 2, 2, 2, 1>;   // 16x8 px x  64 co, 4 waves (2x1), 86 KB: narrow maps (107 / 54 pixels wide)
using X7 = CfgX<16, 1, 8, 2, 2, 2, 1>;   // 16x16 px x 64 co, 4 waves (4x1)
using X8 = CfgX<16, 1, 8, 2, 4, 2, 1>;   // 16x16 px x 64 co, 8 waves (2x1)
using X9 = CfgX<16, 1, 4, 4, 2, 2, 1>;   // 16x8 px x 128 co, 4 waves (2x2)
using X10 = CfgX<32, 1, 8, 4, 4, 2, 1, 1>;  // X1 with the staging loads spread over the MFMA loop
using X11 = CfgX<32, 1, 8, 4, 2, 2, 1, 1>;  // X0 ...
using X12 = CfgX<32, 1, 8, 2, 4, 2, 1, 1>;  // X4 ...
using X13 = CfgX<32, 1, 4, 2, 2, 2, 2, 1>;  // X5 ...
using X14 = CfgX<16, 1, 8, 2, 4, 2, 1, 1>;  // X8 ...
using X15 = CfgX<32, 1, 8, 1, 8, 1, 1, 1>;  // 32x8 px x 32 co, 8 waves (1x1): the skinny outputs (side_prep: 16 couts, input gradient: 3)
using X16 = CfgX<32, 1, 16, 2, 8, 1, 1, 1>; // 32x16 px x 64 co, 8 waves (2x2): twice the pixels per weight byte of X12, a third fewer staging instructions per MFMA than X10
using X17 = CfgX<16, 1, 16, 2, 8, 1, 1, 1>; // 16x32 px x 64 co, 8 waves (2x2): the same for narrow maps
constexpr int kNumTilesX = 18;
template <class C>
constexpr TileInfoX infoX() { return TileInfoX{C::TW, C::TH, C::BN, C::NT, C::LDS_BYTES}; }
const TileInfoX kTilesX[kNumTilesX] = {infoX<X0>(), infoX<X1>(), infoX<X2>(), infoX<X3>(), infoX<X4>(),
                                       infoX<X5>(), infoX<X6>(), infoX<X7>(), infoX<X8>(), infoX<X9>(),
                                       infoX<X10>(), infoX<X11>(), infoX<X12>(), infoX<X13>(), infoX<X14>(), infoX<X15>(), infoX<X16>(), infoX<X17>()};

long tiles_of(const TileInfoX& t, int N, int H, int W, int CoutP) {
  return (long)N * ceil_div(H, t.th) * ceil_div(W, t.tw) * ceil_div(CoutP, t.bn);
}

// Measured on MI355X (tools/tune_x3.py, profiles/r02_tune_x3.txt; 854x480 batch 1): the eight-wave tiles with the staging loads
// spread over the MFMA loop win everywhere -- two waves per SIMD cover each other's vmem-issue and LDS stalls.  128-cout tiles
// (X10: fewest weight bytes per MFMA) when they still give ~one workgroup per CU, else 64-cout tiles; 16 x 16 pixel tiles where
// 32-wide ones pad the frame by more than 10 % (the 107-pixel wide conv4_x); K splits top small grids up to ~200 workgroups.
int pick_tile_x(int N, int H, int W, int CoutP) {
  const bool narrow = (long)ceil_div(W, 32) * 32 * 100 > (long)ceil_div(W, 16) * 16 * 110;
  if (CoutP <= 32) return 15;
  if (CoutP >= 128 && !narrow && tiles_of(kTilesX[10], N, H, W, CoutP) >= 200) return 10;
  return narrow ? 14 : 12;
}

int pick_ksplit_x(const TileInfoX& t, int N, int H, int W, int Cin, int Cout, int CoutP) {
  if (Cin < 256) return 1;
  const long blocks = tiles_of(t, N, H, W, CoutP);
  int ks = 1;
  while (ks < 8 && blocks * ks < 200 && (Cin / 16) / (ks * 2) >= 4) ks *= 2;
  return ks;
}

// pre-split pack: wpk3[((piece * 9 + tap) * CG + cg) * CoutP + co][e] = piece(W[co][8 cg + e][tap])  (dgrad = 0), or the rotated /
// transposed filter of the data gradient (dgrad = 1: roles of Cin and Cout swapped, tap -> 8 - tap); zero padded.
// Every layer's packs in ONE launch (the table rides in the kernel arguments): osvos_net_pack re-packs all 17 filters, forward and
// data-gradient form, after every optimizer step, on the stream the next forward waits on.
// A unit of work = one (channel group cg, 32 output channels m0..m0+31) block of a pack: its 32 x 8 x 9 source values are whole contiguous
// runs of the OIHW filter (forward form: 72 floats per output channel; data-gradient form: 288 floats per reduction channel), read as
// such, turned through LDS, and written as nine 512-byte runs per piece plane.  (Rounds 2-3 gathered one float per thread at a 36-byte
// stride: 15x the filter bytes fetched per launch, 163 us per optimizer step -- profiles/r04_pmc_traffic_configs1.json.)
struct PackX3Table {
  const float* w[OSVOS_PACK_MAX];
  unsigned short* dst[OSVOS_PACK_MAX];
  int Cout[OSVOS_PACK_MAX], Cin[OSVOS_PACK_MAX], dgrad[OSVOS_PACK_MAX];
  unsigned char half[OSVOS_PACK_MAX];  // 1: two FP16 pieces scaled by the filter's block exponent (h2split.h); the filter's largest magnitude sits in word 0 of plane 2
  long start[OSVOS_PACK_MAX + 1];      // in (cg, 32-channel) units
  int n;
};

__device__ inline long pack_plane_elems(const PackX3Table& t, int k) {
  const int K = t.dgrad[k] ? t.Cout[k] : t.Cin[k], M = t.dgrad[k] ? t.Cin[k] : t.Cout[k];
  return 9L * (K / 8) * ((M + 31) / 32 * 32) * 8;
}
// h2 packs, pass 1 and 2: the largest |w| of every filter (bits of a non-negative float: unsigned order = float order)
__global__ void wamax_zero_kernel(PackX3Table t) {
  const int k = (int)threadIdx.x;
  if (k < t.n && t.half[k]) *reinterpret_cast<unsigned*>(t.dst[k] + 2 * pack_plane_elems(t, k)) = 0u;
}
__global__ __launch_bounds__(256) void wamax_kernel(PackX3Table t) {
  const int k = (int)blockIdx.y;
  if (!t.half[k]) return;
  const long len = 9L * t.Cout[k] * t.Cin[k];
  const float* __restrict__ w = t.w[k];
  unsigned m = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < len; i += (long)gridDim.x * 256) {
    const unsigned b = __float_as_uint(w[i]) & 0x7fffffffu;
    m = m > b ? m : b;
  }
  m = h2_wave_max(m);
  if ((threadIdx.x & 63) == 0 && m != 0u) atomicMax(reinterpret_cast<unsigned*>(t.dst[k] + 2 * pack_plane_elems(t, k)), m);
}

__global__ __launch_bounds__(256) void pack_x3_multi_kernel(PackX3Table t) {
  constexpr int ROW = 8 * 9 + 1;                      // one output channel's 8 x 9 values, padded
  __shared__ float tile[32 * ROW];
  for (long blk = blockIdx.x; blk < t.start[t.n]; blk += gridDim.x) {
    int k = 0;
    while (blk >= t.start[k + 1]) ++k;                       // uniform per workgroup: scalar loop
    const int dgrad = t.dgrad[k], Cout = t.Cout[k], Cin = t.Cin[k];
    const int K = dgrad ? Cout : Cin, M = dgrad ? Cin : Cout, MP = (M + 31) / 32 * 32, CG = K / 8;
    const long plane = 9L * CG * MP * 8;
    const int u = (int)(blk - t.start[k]);
    const int cg = u % CG, m0 = (u / CG) * 32;
    const float* __restrict__ w = t.w[k];
    const bool half = t.half[k] != 0;
    const float hsc = half ? h2_pow2(h2_exp(*reinterpret_cast<const unsigned*>(t.dst[k] + 2 * plane))) : 1.f;
    __syncthreads();                                         // the previous unit's reads of `tile`
#pragma unroll
    for (int it = 0; it < 9; ++it) {
      const int L = it * 256 + (int)threadIdx.x;             // 0 .. 2303
      int ml, e, tap;
      float v = 0.f;
      if (dgrad) {                                           // row = reduction channel 8 cg + e (a Cout index): [m0 .. m0 + 31][9] contiguous
        e = L / 288;
        const int j = L % 288;
        ml = j / 9;
        tap = 8 - j % 9;
        if (m0 + ml < M) v = w[((long)(cg * 8 + e) * Cin + m0) * 9 + j];
      } else {                                               // row = output channel m0 + ml: [8 cg .. 8 cg + 7][9] contiguous
        ml = L / 72;
        const int j = L % 72;
        e = j / 9;
        tap = j % 9;
        if (m0 + ml < M) v = w[((long)(m0 + ml) * Cin + cg * 8) * 9 + j];
      }
      tile[ml * ROW + e * 9 + tap] = v;
    }
    __syncthreads();
    const int ml = (int)threadIdx.x >> 3, e = (int)threadIdx.x & 7;
    unsigned short* d = t.dst[k];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      unsigned p0, p1, p2;
      const long i = (((long)tap * CG + cg) * MP + m0 + ml) * 8 + e;
      if (half) {
        h2_split2(tile[ml * ROW + e * 9 + tap], 0.f, hsc, p0, p1);
        d[i] = (unsigned short)(p0 & 0xffffu);
        d[plane + i] = (unsigned short)(p1 & 0xffffu);
        continue;
      }
      split2(tile[ml * ROW + e * 9 + tap], 0.f, p0, p1, p2);
      d[i] = (unsigned short)(p0 & 0xffffu);
      d[plane + i] = (unsigned short)(p1 & 0xffffu);
      d[2 * plane + i] = (unsigned short)(p2 & 0xffffu);
    }
  }
}

}  // namespace

// n packs (n <= OSVOS_PACK_MAX) in one launch: ws[k] OIHW fp32 [Couts[k]][Cins[k]][3][3] -> dsts[k] (osvos_pack_x3 layout; dgrads[k] != 0: data-gradient form)
int osvos_pack_x3_multi(const float* const* ws, void* const* dsts, const int* Couts, const int* Cins, const int* dgrads, int n, hipStream_t stream) {
  return osvos_pack_x3_multi_fmt(ws, dsts, Couts, Cins, dgrads, nullptr, n, stream);
}
// halfs[k] != 0: entry k in the two-piece FP16 format (h2split.h; three launches instead of one: zero + largest magnitude + pack); NULL: every
// entry in the format of this thread's osvos_x3_pieces() (22 = FP16 pairs, else three bf16 planes)
int osvos_pack_x3_multi_fmt(const float* const* ws, void* const* dsts, const int* Couts, const int* Cins, const int* dgrads, const int* halfs, int n,
                            hipStream_t stream) {
  OSVOS_ARG_CHECK(ws && dsts && Couts && Cins && dgrads && n >= 0 && n <= OSVOS_PACK_MAX, "pack_x3_multi: bad table (n = %d)", n);
  if (n == 0) return 0;
  PackX3Table t;
  t.n = n;
  t.start[0] = 0;
  bool any_half = false;
  for (int k = 0; k < n; ++k) {
    t.half[k] = (halfs != nullptr ? halfs[k] != 0 : osvos_x3_pieces() == 22) ? 1 : 0;
    any_half = any_half || t.half[k];
    const int K = dgrads[k] ? Couts[k] : Cins[k], M = dgrads[k] ? Cins[k] : Couts[k];
    OSVOS_ARG_CHECK(ws[k] && dsts[k] && K % 16 == 0 && M > 0, "pack_x3_multi: entry %d (K = %d, M = %d)", k, K, M);
    t.w[k] = ws[k]; t.dst[k] = reinterpret_cast<unsigned short*>(dsts[k]); t.Cout[k] = Couts[k]; t.Cin[k] = Cins[k]; t.dgrad[k] = dgrads[k] ? 1 : 0;
    t.start[k + 1] = t.start[k] + (long)(K / 8) * (osvos_cout_pad(M) / 32);
  }
  if (any_half) {
    hipLaunchKernelGGL(wamax_zero_kernel, dim3(1), dim3(64), 0, stream, t);
    hipLaunchKernelGGL(wamax_kernel, dim3(64, (unsigned)n), dim3(256), 0, stream, t);
  }
  const long blocks = t.start[n] < 8192 ? t.start[n] : 8192;
  hipLaunchKernelGGL(pack_x3_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, t);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

// bytes of the pre-split pack of a conv with `K` reduction channels (multiple of 16) and `M` output channels
size_t osvos_wpack_x3_bytes(int M, int K) { return (size_t)3 * 9 * K * osvos_cout_pad(M) * 2; }

// w: OIHW fp32 [Cout][Cin][3][3].  dgrad = 0: pack for the forward conv (K = Cin, M = Cout); 1: for the data gradient (K = Cout, M = Cin)
int osvos_pack_x3(const float* w, void* wpk3, int Cout, int Cin, int dgrad, hipStream_t stream) {
  OSVOS_ARG_CHECK(w && wpk3 && Cout > 0 && Cin > 0, "pack_x3: bad arguments");
  const int K = dgrad ? Cout : Cin;
  OSVOS_ARG_CHECK(K % 16 == 0, "pack_x3: %d reduction channels (must be a multiple of 16)", K);
  const float* ws[1] = {w};
  void* dsts[1] = {wpk3};
  const int co[1] = {Cout}, ci[1] = {Cin}, dg[1] = {dgrad};
  return osvos_pack_x3_multi(ws, dsts, co, ci, dg, 1, stream);
}

int osvos_conv3x3_f32x3_num_tiles(void) { return kNumTilesX; }

// stream-K workspace (ConvEpi::sk_ws): tickets (must be ZERO before the first launch that uses the buffer; every launch leaves them zero) +
// two partial slots per persistent workgroup.  Independent of the layer's shape; launches that share it must be stream-ordered.
size_t osvos_conv3x3_f32x3_streamk_ws_bytes(void) { return kSkTicketBytes + (size_t)kSkMaxGrid * 2 * kSkSlotBytes; }
size_t osvos_conv3x3_f32x3_streamk_ticket_bytes(void) { return kSkTicketBytes; }

// Cout may be ragged (the 3-channel input gradient) as long as the output has room for the rounded-up channel quad: the pack's
// padded couts carry zero weights, so the extra channel is written as 0
bool osvos_conv3x3_f32x3_applicable(int Cin, int Cout, int y_cs) { return Cin % 16 == 0 && y_cs % 4 == 0 && ((Cout + 3) & ~3) <= y_cs; }

// same contract as osvos_conv3x3_f32_ws (conv3x3_f32.hip); tile: -1 = automatic, 0..kNumTilesX-1 (+100: XCD-local halo map)
int osvos_conv3x3_f32x3(const float* x, const float* wpk, const float* bias, const float* mask, float* y,
                        int N, int H, int W, int Cin, int Cout, int y_cs, int relu, int tile, int ksplit, void* part_ws, hipStream_t stream) {
  return osvos_conv3x3_f32x3_ps(x, wpk, nullptr, bias, mask, y, N, H, W, Cin, Cout, y_cs, relu, tile, ksplit, part_ws, stream);
}

// wpk3 != NULL: pre-split pack (osvos_pack_x3) -- wpk (the fp32 pack) is then not read and may be NULL
int osvos_conv3x3_f32x3_ps(const float* x, const float* wpk, const void* wpk3, const float* bias, const float* mask, float* y,
                           int N, int H, int W, int Cin, int Cout, int y_cs, int relu, int tile, int ksplit, void* part_ws, hipStream_t stream) {
  return osvos_conv3x3_f32x3_epi(x, wpk, wpk3, bias, mask, y, N, H, W, Cin, Cout, y_cs, relu, tile, ksplit, part_ws, nullptr, stream);
}

// epi (may be NULL): fused epilogues (epi.h).  epi->pooled: the launch needs one of the eight-wave tiles whose waves hold whole 2 x 2
// windows (10, 12, 14) and is never cut along K by partial-sum launches (the stream-K form keeps whole tiles in one workgroup's registers)
int osvos_conv3x3_f32x3_epi(const float* x, const float* wpk, const void* wpk3, const float* bias, const float* mask, float* y,
                            int N, int H, int W, int Cin, int Cout, int y_cs, int relu, int tile, int ksplit, void* part_ws, const ConvEpi* epi,
                            hipStream_t stream) {
  const bool pool_fwd = epi != nullptr && epi->pooled != nullptr;
  OSVOS_ARG_CHECK(x && (wpk || wpk3) && y, "conv3x3 f32x3: null pointer");
  OSVOS_ARG_CHECK(!pool_fwd || (relu && Cout % 4 == 0 && y_cs == Cout), "conv3x3 f32x3: fused pool forward needs ReLU and a dense Cout %% 4 == 0 result");
  OSVOS_ARG_CHECK(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv3x3 f32x3: bad shape");
  OSVOS_ARG_CHECK(osvos_conv3x3_f32x3_applicable(Cin, Cout, y_cs), "conv3x3 f32x3: needs Cin %% 16 == 0 (%d), y_cs %% 4 == 0 and >= Cout rounded up to 4 (%d, %d)",
                  Cin, Cout, y_cs);
  OSVOS_ARG_CHECK(Cout % 4 == 0 || (bias == nullptr && mask == nullptr), "conv3x3 f32x3: ragged Cout (%d) takes no bias / mask", Cout);
  OSVOS_ARG_CHECK(y_cs >= Cout, "conv3x3 f32x3: y channel stride %d < Cout %d", y_cs, Cout);
  OSVOS_ARG_CHECK((long)H * W * Cin < (1L << 29) && (long)H * W * y_cs < (1L << 29), "conv3x3 f32x3: image too large for 31-bit byte offsets");
  ConvArgsX a;
  a.x = x; a.wpk = wpk; a.wpk3 = reinterpret_cast<const uint4*>(wpk3); a.bias = bias; a.mask = mask; a.y = y;
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = (Cout + 3) & ~3; a.CoutP = osvos_cout_pad(Cout); a.y_cs = y_cs;
  a.relu = relu;
  if (epi != nullptr) a.epi = *epi;
  if (tile < 0) {
    OSVOS_ENV_INT(env_tile, "OSVOS_X3_TILE", -1);
    tile = env_tile >= 0 ? env_tile : pick_tile_x(N, H, W, a.CoutP);
    // tuning knobs for the two-piece forms (half the matrix work per staged byte): another tile where the rule says 10 / 12 (launches without the
    // fused pool only: that epilogue exists for tiles 10, 12, 14)
    OSVOS_ENV_INT(wide2, "OSVOS_X2_TILE_FOR_10", 10);
    OSVOS_ENV_INT(mid2, "OSVOS_X2_TILE_FOR_12", 12);
    if (env_tile < 0 && osvos_x3_pieces() != 3 && !pool_fwd) tile = tile == 10 ? wide2 : (tile == 12 ? mid2 : tile);
    // XCD-local map (cout tiles of one spatial tile on one XCD) only where the activations are MUCH larger than the weights: the rule of rounds 2-4
    // (pixels > 9 CoutP, i.e. fp32 activation bytes > fp32 weight bytes) put conv4_x on it, where it measures 6-7 % slower per launch than the plain
    // order (X14 on conv4_2: 0.166 vs 0.155 ms; the pre-split pack is 1.5x the fp32 weights and every XCD then streams all of it) -- with the factor 3
    // the headline step gains 1.0-2.8 % on three boxes, configs[4] and the window-fused form are level (profiles/r05_ab_x3_map_rule.txt)
    if (env_tile < 0 && (double)H * W * Cin > 27.0 * Cin * a.CoutP) tile += 100;
  }
  a.map = tile >= 100 ? 1 : 0;
  tile %= 100;
  OSVOS_ARG_CHECK(tile >= 0 && tile < kNumTilesX, "conv3x3 f32x3: unknown tile config %d", tile);
  a.part = reinterpret_cast<float*>(part_ws);
  a.ksplit = 1;
  if (part_ws != nullptr) {
    OSVOS_ENV_INT(env_ks, "OSVOS_X3_KSPLIT", 0);
    a.ksplit = ksplit > 0 ? ksplit : (env_ks > 0 && Cin >= 256 ? env_ks : pick_ksplit_x(kTilesX[tile], N, H, W, Cin, Cout, a.CoutP));
    if (a.ksplit < 1 || a.ksplit > 8 || a.ksplit > (Cin >> 4) || Cout % 4 != 0) a.ksplit = 1;
  }
  if (epi != nullptr && (epi->mask_bits != nullptr || epi->y_bits != nullptr))
    OSVOS_ARG_CHECK(Cout % 32 == 0 && y_cs == Cout, "conv3x3 f32x3: one-bit masks need a dense result with Cout %% 32 == 0 (Cout %d, stride %d)", Cout, y_cs);
  if (epi != nullptr && epi->y_bits != nullptr) a.ksplit = 1;
  if (pool_fwd) {
    a.ksplit = 1;
    OSVOS_ARG_CHECK(tile == 10 || tile == 12 || tile == 14, "conv3x3 f32x3: fused pool forward is built for tiles 10, 12 and 14 (got %d)", tile);
  }
  // stream-K (kernel header): taken when the caller hands a workspace and the plain grid would leave a sizeable part of the chip without a
  // tile -- one workgroup per CU, so a grid of `ntiles` runs in ceil(ntiles / CUs) rounds and loses 1 - ntiles / (rounds x CUs) of them
  // (conv3_x at 854x480 batch 1: 210 tiles, 18 %) -- or would need partial-sum launches + a finalize kernel to fill it (conv5_x)
  int sk_grid = 0;
  a.sk_order = 0; a.sk_tickets = nullptr; a.sk_part = nullptr;
  if (epi != nullptr && epi->sk_ws != nullptr && wpk3 != nullptr && (tile == 10 || tile == 12 || tile == 14) && Cout % 4 == 0) {
    const long ntiles = tiles_of(kTilesX[tile], N, H, W, a.CoutP), units = ntiles * (Cin >> 4);
    OSVOS_ENV_INT(env_grid, "OSVOS_X3_STREAMK_GRID", 0);            // tuning: persistent workgroups (default: the CU count)
    OSVOS_ENV_INT(env_loss, "OSVOS_X3_STREAMK_MIN_LOSS", 6);        // tuning: percent of the chip a plain grid must leave idle
    int g = epi->sk_grid > 0 ? epi->sk_grid : (env_grid > 0 ? env_grid : osvos_cu_count());
    if (g > kSkMaxGrid) g = kSkMaxGrid;
    const long rounds = (ntiles * a.ksplit + g - 1) / g;
    // Measured per layer at 854x480 batch 1 (profiles/r04_tune_streamk.txt, three boxes): the 64-cout tiles (12, 14: 64 KB partial slots) WIN --
    // conv1_2 -6 %, conv4_1 -4 %, conv4_2 / 4_3 -5...-7 %, conv5_x level with partial-sum launches + finalize and one launch fewer -- while the
    // 128-cout tile (10: conv2_x / conv3_x, 128 KB slots, short K ranges on conv2_x / conv3_1) LOSES 1-12 % although every workgroup issues a
    // fifth fewer MFMAs.  The automatic choice therefore covers tiles 12 and 14 only; tile 10 runs stream-K when forced (tests, tuning).
    const bool lossy = (tile == 12 || tile == 14) && (a.ksplit > 1 || (rounds * g - ntiles) * 100 >= (long)env_loss * rounds * g);
    if (ntiles <= kSkMaxTiles && units >= g && (epi->sk_grid > 0 || lossy)) {
      sk_grid = g;
      a.ksplit = 1;
      a.sk_order = a.map ? 0 : 1;      // activations > weights: Cout tiles of one halo back to back; else one weight slice per XCD
      OSVOS_ENV_INT(env_order, "OSVOS_X3_STREAMK_ORDER", -1);
      if (env_order == 0 || env_order == 1) a.sk_order = env_order;
      a.sk_tickets = reinterpret_cast<unsigned*>(epi->sk_ws);
      a.sk_part = reinterpret_cast<float*>(reinterpret_cast<char*>(epi->sk_ws) + kSkTicketBytes);
    }
  }
  int rc;
  switch (tile) {
    case 0: rc = launch_x<X0>(a, 0, stream); break;
    case 1: rc = launch_x<X1>(a, 0, stream); break;
    case 2: rc = launch_x<X2>(a, 0, stream); break;
    case 3: rc = launch_x<X3>(a, 0, stream); break;
    case 4: rc = launch_x<X4>(a, 0, stream); break;
    case 5: rc = launch_x<X5>(a, 0, stream); break;
    case 6: rc = launch_x<X6>(a, 0, stream); break;
    case 7: rc = launch_x<X7>(a, 0, stream); break;
    case 8: rc = launch_x<X8>(a, 0, stream); break;
    case 9: rc = launch_x<X9>(a, 0, stream); break;
    case 10: rc = launch_x<X10, true>(a, sk_grid, stream); break;
    case 11: rc = launch_x<X11>(a, 0, stream); break;
    case 12: rc = launch_x<X12, true>(a, sk_grid, stream); break;
    case 13: rc = launch_x<X13>(a, 0, stream); break;
    case 14: rc = launch_x<X14, true>(a, sk_grid, stream); break;
    case 15: rc = launch_x<X15>(a, 0, stream); break;
    case 16: rc = launch_x<X16>(a, 0, stream); break;
    case 17: rc = launch_x<X17>(a, 0, stream); break;
    default: osvos_set_error("conv3x3 f32x3: unknown tile config %d", tile); return -1;
  }
  if (rc) return rc;
  return a.ksplit > 1 ? osvos_conv3x3_splitk_finalize_f32(a.part, bias, mask, y, (long)N * H * W, Cout, y_cs, a.ksplit, relu, stream) : 0;
}
