// 3x3 convolution, Cin = 64, bf16 activations in and out (bf16-store mode): persistent workgroups, resident filter, DEFERRED + SKEWED epilogue.
// (round 6; VERDICT r05 item 3.  Layers: conv1_2, conv2_1 forward and conv1_2's data gradient -- vgg_osvos.py:136-145 and their autograd.)
//
// What bounds these launches (profiles/r06_p64_diagnosis.txt): K = 576 is FOUR 16-channel chunks, so per output element the matrix pipe has the
// least work of the whole net, while the epilogue (bias, ReLU, bf16 rounding, sign bits, 2x2 pool + code bytes, 16-byte stores) costs the same
// per element everywhere.  In the register-staged tile 9 and in the LDS-DMA kernel with a resident filter (tile 36) alike the waves of a workgroup
// run their epilogues TOGETHER, right after the last chunk's barrier: 9 VALU instructions per MFMA, MFMA busy 33-39 %, VALU time per SIMD (190 us)
// above MFMA time (169 us), and the two never overlap.  Making the filter resident alone (tile 36) changed nothing: 0.58 vs 0.55 ms.
//
// This kernel therefore
//   * keeps the resident-filter persistent structure of conv3x3_bf16_dma.hip (one workgroup per CU, 8 waves, 512 px x 64 couts per tile, the
//     73.7 KB filter loaded once, four rotating 21.5 KB activation buffers filled by `buffer_load ... lds` three chunks ahead, one barrier per
//     chunk);
//   * DEFERS a tile's epilogue: at the tile's end the accumulators (+ bias) move to a pending register set, and the four row-pieces of the
//     epilogue run inside the four chunk iterations of the NEXT tile;
//   * SKEWS the two waves that share a SIMD: waves 0-3 run [MFMAs of the chunk][epilogue piece], waves 4-7 [epilogue piece][MFMAs] -- while one
//     wave of a SIMD holds the matrix pipe the other one does its VALU work and stores;
//   * does the epilogue on PACKED bf16 pairs (ReLU = v_pk_max_i16 against 0, sign bits from v_pk_min_u16, pool / first-maximum code bytes with
//     v_pk_max_u16 / v_pk_mad_u16): ~300-500 VALU instructions per wave and tile instead of ~1300-1700;
//   * counts `s_waitcnt vmcnt` EXACTLY (stores of the epilogue pieces may stay in flight across barriers; only the activation chunk that is
//     needed next has to have landed).
// Results are bit-identical to the other tiles' epilogues (same fp32 bias add, same RNE rounding; ReLU and rounding commute).
#include <type_traits>

#include "common.h"
#include "maskbits.h"
#include "kernels.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct P64Args {
  const bf16_t* x;       // bf16 NHWC, channel stride 64
  const uint4* wpk;      // bf16 pack [9][8][CoutP][8]
  const float* bias;     // may be NULL
  bf16_t* ybf;
  int N, H, W, Cout, CoutP, y_cs;
  int tiles_x, tiles_y, nct, nsp, map, ntiles, band;
  int relu;
  const unsigned* mask_bits;   // MODE 2: one-bit-per-element ReLU mask of a data gradient (maskbits.h)
  unsigned* y_bits;            // MODE 0, optional: sign bits of the result (needs relu)
  bf16_t* pooled;              // MODE 1: maxpool2x2 (ceil mode) of the result
  unsigned char* pool_code;    // MODE 1, optional: code bytes (pool.hip)
  unsigned long long* prof;    // phase cycle sums per wave (read only by probe builds, -DP64_PROF; tools/p64_phase_probe.py)
};

constexpr int TW = 32, HWD = TW + 2, TH = 16, HHT = TH + 2, PLANE = HHT * HWD;     // 512 px tile, halo 18 x 34
constexpr int KG = 2, CIN = 64, NCH = CIN / 16, RCG = CIN / 8, BN = 64;
constexpr int NW = 8, NT = 64 * NW;
constexpr int A_SLOTS = (KG * PLANE + 63) / 64 * 64, A_INSTR = A_SLOTS / 64;          // 1280 slots, 20 DMA instructions per chunk
constexpr int NBUF = 4, DIST = NBUF - 1;                                             // chunk g lives in buffer g % 4 = its index inside the tile
constexpr int BUF_SLOTS = A_SLOTS + 64;                                              // + one spare instruction target
constexpr int NA = (A_INSTR + NW - 1) / NW;                                          // DMA instructions per wave and chunk (3; some are spares)
constexpr int RB_BASE = NBUF * BUF_SLOTS, RB_SLOTS = 9 * RCG * BN, RB_INSTR = RB_SLOTS / 64;
constexpr int BIAS_BASE = RB_BASE + RB_SLOTS;                                        // 64 fp32 bias values of the workgroup's cout tile (16 slots)
constexpr size_t LDS_BYTES = (size_t)(BIAS_BASE + 16) * 16;
constexpr unsigned OOB = 0x80000000u;
static_assert(NCH == NBUF, "a chunk's buffer is its index inside the tile");
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

// vector-memory instructions of epilogue piece kc (every lane issues all of them; absent tensors / pixels outside the image go out of range)
//   MODE 0 (forward: bias, ReLU, sign bits): 2 result stores + 1 bit-word store
//   MODE 1 (forward + fused pool): 2 result stores; pieces 1 and 3 also 2 pooled + 2 code stores
//   MODE 2 (data gradient with a one-bit mask): 2 result stores; the four mask words are loaded when the tile becomes pending (E = 4)
constexpr int piece_ops(int mode, int kc) { return mode == 0 ? 3 : (mode == 1 ? ((kc & 1) ? 6 : 2) : 2); }
constexpr int tile_end_ops(int mode) { return mode == 2 ? 4 : 0; }
// vmcnt that may stay outstanding at the wait of chunk iteration kc so that the DMA of stream chunk (this + 1) -- issued DIST - 1 = 2 iterations
// ago -- is complete.  grp 0: an iteration issues [DMA x NA][piece], grp 1: [piece][DMA x NA]; tile-end loads follow iteration 3.
// phase 0: the workgroup's first tile (nothing pending: no pieces); 1: its second tile (the two iterations before it had no pieces); 2: steady state
constexpr int allowed_vm(int mode, int grp, int kc, int phase) {
  int n = 0;
  for (int back = 2; back >= 0; --back) {            // iterations kc - 2, kc - 1, kc
    const int j = kc - back, jk = (j + 4) & 3;
    const bool pieces = phase == 2 || (phase == 1 && j >= 0);
    const int s = pieces ? piece_ops(mode, jk) : 0;
    const int e = (jk == 3 && back > 0 && phase >= 1) ? tile_end_ops(mode) : 0;      // issued after iteration 3 (a previous tile's: it exists from phase 1 on)
    if (back == 2) n += (grp == 0 ? s : 0) + e;      // grp 0: the piece of that iteration came after its DMA; grp 1: the DMA was its last instruction
    else n += NA + s + e;
  }
  return n;
}
static_assert(allowed_vm(0, 0, 0, 0) == 2 * NA && allowed_vm(0, 1, 2, 0) == 2 * NA, "first tile: only the two younger chunks' DMA");
static_assert(allowed_vm(1, 0, 3, 2) <= 63 && allowed_vm(2, 0, 1, 2) <= 63, "vmcnt is six bits");

#ifndef P64_DBG
#define P64_DBG 0
#endif
template <int VM>
__device__ inline void wait_vm() {
  if (P64_DBG & 2) __builtin_amdgcn_s_waitcnt(0x0F70);
  else __builtin_amdgcn_s_waitcnt(0x0F70 | (VM & 15) | ((VM >> 4) << 14));
}

__device__ inline unsigned dpp_xor1(unsigned v) {      // value of lane ^ 1 (quad_perm [1, 0, 3, 2])
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
}
// Packed 16-bit ops through (non-volatile) asm: written as vector C, hipcc rewrites min(x, 1) into (x != 0) ? 1 : 0 and then scalarises the whole
// chain into per-half v_cmp / v_cndmask / v_perm (MODE 1 came out at 3,400 VALU instructions: 800 v_cndmask, 360 v_perm, 670 v_cmp).
__device__ inline unsigned pk_max_u16(unsigned a, unsigned b) {
  unsigned r;
  asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ inline unsigned pk_min_u16(unsigned a, unsigned b) {
  unsigned r;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ inline unsigned pk_sub(unsigned a, unsigned b) {
  unsigned r;
  asm("v_pk_sub_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ inline unsigned pk_mad(unsigned a, unsigned b, unsigned c) {      // a * b + c per half
  unsigned r;
  asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ inline unsigned pk_relu(unsigned a, unsigned zero) {              // max(int16 pattern, 0) per half
  unsigned r;
  asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(zero));
  return r;
}

template <int MODE>
__global__ __launch_bounds__(NT, 2) void conv3x3_bf16_p64_kernel(P64Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint4* lds = reinterpret_cast<const uint4*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  // waves w and w + 4 share a SIMD: group 0 = multiply first, 1 = epilogue piece first.  The value comes out of an asm statement with a scalar
  // output: hipcc folds readfirstlane(tid >> 6) back to tid >> 6, calls the branch on it divergent and then keeps the DMA descriptors that are
  // live across it in VECTOR registers -- which `buffer_load ... lds` cannot take
  int grp;
  asm volatile("s_lshr_b32 %0, %1, 2" : "=s"(grp) : "s"(wv) : "scc");
  if (P64_DBG & 1) grp = 0;
  const unsigned K1 = 0x00010001u, K2 = 0x00020002u, K4 = 0x00040004u, K0 = 0u;      // per-half constants of the packed epilogue (VGPR operands)

  // A workgroup's tiles are t = blockIdx.x, + G, + 2 G, ...: the cout tile never changes (G is a multiple of 8 nct) and the SPATIAL index advances by
  // the constant D = G / nct, so a tile cursor (image, tile row, tile column) is decoded ONCE (the only integer divisions of the kernel) and then
  // stepped by the mixed-radix digits of D with two carries -- a dozen scalar instructions per tile instead of six runtime divisions (the probe
  // showed 2,400-3,400 cycles per tile in the two decodes: profiles/r06_p64_diagnosis.txt).
  struct Tile { int n, ty, tx; };               // live <=> n < N
  int co0;
  Tile first;
  {
    const int t = blockIdx.x;
    int sp, ct;
    if (a.map == 0) {
      sp = t / a.nct;
      ct = t % a.nct;
    } else if (a.map == 1) {
      const int j = t >> 3;
      ct = j % a.nct;
      sp = (j / a.nct) * 8 + (t & 7);
    } else {
      // map 2, XCD bands: block b runs on XCD b % 8 and XCD k owns the CONTIGUOUS spatial tiles [k per, (k + 1) per): the 32 workgroups of an XCD walk
      // 32 raster-consecutive tiles at a time, so the halo columns of x-neighbours and the halo rows of the tile row above are hits in THAT XCD's L2
      // (map 1 deals consecutive tiles to different XCDs: FETCH_SIZE 994 MB for 630 MB of input, profiles/r06_p64_diagnosis.txt)
      const int j = t >> 3;
      ct = j % a.nct;
      sp = (t & 7) * a.band + j / a.nct;      // (tiles past a band's end: the launcher sizes the walk by the band, the last band's tail is dead: n >= N)
    }
    co0 = ct * BN;
    first.tx = sp % a.tiles_x;
    sp /= a.tiles_x;
    first.ty = sp % a.tiles_y;
    first.n = sp / a.tiles_y;
  }
  int dn, dy, dx;                               // digits of the spatial stride D
  {
    int D = (int)gridDim.x / a.nct / (a.map == 2 ? 8 : 1);
    dx = D % a.tiles_x;
    D /= a.tiles_x;
    dy = D % a.tiles_y;
    dn = D / a.tiles_y;
  }
  auto step = [&](Tile& T) __attribute__((always_inline)) {
    T.tx += dx;
    int c = T.tx >= a.tiles_x ? 1 : 0;
    T.tx -= c ? a.tiles_x : 0;
    T.ty += dy + c;
    c = T.ty >= a.tiles_y ? 1 : 0;
    T.ty -= c ? a.tiles_y : 0;
    T.n += dn + c;
  };
  int my_tiles = 0;
  for (int t = blockIdx.x; t < a.ntiles; t += gridDim.x) ++my_tiles;      // (map 2: ntiles = 8 nct band -- every XCD walks its whole band)

  auto make_rsrc = [](const void* p, int bytes) __attribute__((always_inline)) -> i32x4 {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    return i32x4{(int)(unsigned)v, (int)(unsigned)(v >> 32), bytes, 0x00020000};
  };
  const size_t ximg_elems = (size_t)a.H * a.W * CIN;
  const i32x4 wrs = make_rsrc(a.wpk, (int)((size_t)9 * RCG * a.CoutP * 16));
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  auto dma16 = [](const i32x4& rs, unsigned lds_addr, unsigned voff, int soff) __attribute__((always_inline)) {
    if (P64_DBG & 8) asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_nop 4" : : "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "m0", "memory");
    else asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "m0", "memory");
  };
#pragma clang diagnostic pop

  // ---- issue side: the DMA stream runs DIST chunks ahead of the multiply side and crosses tile boundaries on its own
  unsigned a_off[NA];
  int a_hyx[NA];                                // this lane's halo slot of instruction i: hy << 8 | hx, bit 30 = channel group, bit 31 = no slot (tile-invariant)
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int e = 64 * (wave + NW * i) + lane;
    const int g = e / PLANE, rem = e % PLANE;
    a_hyx[i] = (e < KG * PLANE) ? ((rem / HWD) << 8 | (rem % HWD) | (g << 30)) : (int)0x80000000;
  }
  i32x4 xrs = make_rsrc(a.x, 0);
  auto set_issue_tile = [&](const Tile& T) __attribute__((always_inline)) {
    const bool live = T.n < a.N;
    xrs = make_rsrc(a.x + (size_t)(live ? T.n : 0) * ximg_elems, live ? (int)(ximg_elems * 2) : 0);      // (a dead tile's descriptor has no records: zeros land)
    const int y0 = T.ty * TH - 1, x0 = T.tx * TW - 1;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int gy = y0 + ((a_hyx[i] >> 8) & 0xff), gx = x0 + (a_hyx[i] & 0xff), g = (a_hyx[i] >> 30) & 1;
      a_off[i] = (a_hyx[i] >= 0 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? (unsigned)(((gy * a.W + gx) * CIN + 8 * g) * 2) : OOB;
    }
  };
  auto dma_one = [&](int d, int kc, int buf, unsigned dead) __attribute__((always_inline)) {      // d-th instruction of this wave for chunk kc into buffer buf
    const unsigned base = lds0 + (unsigned)(buf * BUF_SLOTS * 16);
    const int j = wv + NW * d;
    dma16(xrs, base + (unsigned)(j < A_INSTR ? j * 1024 : A_SLOTS * 16), a_off[d] | dead, kc * (8 * KG * 2));
  };

  const int total = my_tiles * NCH;
  int ikc = 0, gi = 0;
  Tile icur = first;                            // issue-side tile cursor
  {
    // the whole filter of this workgroup's cout tile, once (the launcher keeps gridDim.x a multiple of 8 nct: every tile walked has the same one)
    for (int j = wv; j < RB_INSTR; j += NW) {
      const int e = 64 * j + lane;
      const int tap = e / (RCG * BN), rem = e % (RCG * BN);
      const int g = rem / BN, nn = rem % BN;
      dma16(wrs, lds0 + (unsigned)((RB_BASE + 64 * j) * 16), co0 + nn < a.CoutP ? (unsigned)(((tap * RCG + g) * a.CoutP + co0 + nn) * 16) : OOB, 0);
    }
  }
  set_issue_tile(icur);
  auto issue_advance = [&]() __attribute__((always_inline)) {
    ++gi;
    if (++ikc == NCH) {
      ikc = 0;
      step(icur);
      if (gi < total) set_issue_tile(icur);
    }
  };
#pragma unroll
  for (int c = 0; c < DIST; ++c) {
#pragma unroll
    for (int d = 0; d < NA; ++d) dma_one(d, ikc, c, gi < total ? 0u : OOB);
    issue_advance();
  }
  wait_vm<(DIST - 1) * NA>();                   // the filter and chunk 0 have landed (this wave's part) ...
  __syncthreads();                              // ... everybody's

  // ---- multiply side
  const int a_idx = lh * PLANE + (wm * 4) * HWD + li;                     // + (halo row 0..5 of the wave) * HWD + tap column
  const int rb_idx = RB_BASE + lh * BN + wn * 32 + li;                    // + (tap * RCG + kc * KG) * BN
  f32x16 acc[4];
  unsigned ph[4][8];                            // the pending tile, bias added and rounded to bf16 pairs: ph[mi][2 q] = couts (e 0, 1), [2 q + 1] = (e 2, 3) of quad q
  Tile P = first;                               // the pending tile (its epilogue pieces run inside the next tile's chunk iterations)
  bool have_pend = false;
  unsigned mw[4] = {0u, 0u, 0u, 0u};            // MODE 2: mask words of the pending tile's four rows
  u32x4 keep[2][2];                             // MODE 1: packed results of the row pair being pooled

  // bias of the workgroup's 64 couts -> LDS (the same for every tile; read back when a tile retires: registers are the scarce resource here)
  if (tid < 16) {
    const int cb = co0 + 4 * tid;
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(a.bias != nullptr ? (void*)const_cast<float*>(a.bias) : (void*)const_cast<uint4*>(a.wpk), 0,
                                                                         a.bias != nullptr ? a.Cout * 4 : 0, 0x00020000);
    reinterpret_cast<u32x4*>(smem)[BIAS_BASE + tid] = __builtin_amdgcn_raw_buffer_load_b128(brs, cb * 4, 0, 0);
  }
  // (visible to every wave after the first chunk iteration's barrier; the first read happens when the first tile retires, four barriers later)

  // one chunk's 36 MFMAs; the NA DMA instructions of stream chunk (this + DIST) are issued between them.
  // Fragment traffic: the A fragment of (row mi, tap row r, tap column s) is halo row mi + r at column offset s -- the SAME registers for every
  // (mi, r) with the same mi + r.  Per tap column the wave reads its 6 halo rows once and uses them for 12 MFMAs: 18 A + 9 B fragment reads per chunk
  // instead of 36 + 9.  (With one wave per SIMD in its multiply phase at a time -- the other one is in its epilogue piece -- nothing else hides LDS
  // latency: at 45 reads per 36 MFMAs the four multiplying waves asked 160 B/clk of the LDS pipe and the MFMAs ran at half rate,
  // profiles/r06_p64_diagnosis.txt.)
  auto mfma_chunk = [&](auto KC_) __attribute__((always_inline)) {
    constexpr int KC = decltype(KC_)::value;
    constexpr int tgt = (KC + DIST) % NBUF;
    const unsigned dead = (gi < total && !(P64_DBG & 128)) ? 0u : OOB;
    const int dkc = ikc;
    const uint4* As = lds + (size_t)KC * BUF_SLOTS;
    // 18 row steps k = 6 sc + rho: halo row rho at tap column sc serves the (mi, r) pairs with mi + r = rho -- 1, 2, 3, 3, 2, 1 MFMAs.  A fragments
    // live in a ring of RING registers sets (requested RING - 1 steps ahead), the three B fragments of a tap column are double-buffered.
    constexpr int RING = 5;
    uint4 fa[RING], fb[2][3];
    auto lda = [&](int k) __attribute__((always_inline)) { fa[k % RING] = As[a_idx + (k % 6) * HWD + k / 6]; };
    auto ldb = [&](int sc) __attribute__((always_inline)) {
#pragma unroll
      for (int r = 0; r < 3; ++r) fb[sc & 1][r] = lds[rb_idx + ((r * 3 + sc) * RCG + KC * KG) * BN];
    };
    ldb(0);
#pragma unroll
    for (int k = 0; k < RING - 1; ++k) lda(k);
#pragma unroll
    for (int k = 0; k < 18; ++k) {
      const int sc = k / 6, rho = k % 6;
      if (k + RING - 1 < 18) lda(k + RING - 1);
      if (rho == 1 && sc + 1 < 3) ldb(sc + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int r = rho - mi;
        if (r < 0 || r > 2) continue;
        if (KC == 0 && sc == 0 && r == 0) {         // first product of the tile: C = 0 (an inline constant: no zeroing pass)
          const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb[sc & 1][r]), __builtin_bit_cast(bf16x8_t, fa[k % RING]), z, 0, 0, 0);
        } else {
          acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb[sc & 1][r]), __builtin_bit_cast(bf16x8_t, fa[k % RING]), acc[mi], 0, 0, 0);
        }
      }
      if (rho == 2 && sc < NA) dma_one(sc, dkc, tgt, dead);      // one DMA instruction per tap column
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  static_assert(NA <= 3, "one DMA instruction per tap column");

  // ---- epilogue piece MI of the pending tile: row wm * 4 + MI of the wave's 32 pixels x 32 couts
  const bool pool_code = MODE == 1 && __builtin_amdgcn_readfirstlane((int)(a.pool_code != nullptr)) != 0;      // (pinned to a scalar: left to itself hipcc re-derives the comparison in VECTOR registers at every use and drags the DMA descriptors with it)
  const int PHo = (a.H + 1) / 2, PWo = (a.W + 1) / 2;
  auto piece = [&](auto MI_) __attribute__((always_inline)) {
    constexpr int MI = decltype(MI_)::value;
    const size_t img_elems = (size_t)a.H * a.W * a.y_cs;
    void* const anyp = const_cast<uint4*>(a.wpk);
    const bool live = P.n < a.N;
    const int pn = live ? P.n : 0;
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.ybf + pn * img_elems), 0, (live && !(P64_DBG & 32)) ? (int)(img_elems * 2) : 0, 0x00020000);
    const int oy = P.ty * TH + wm * 4 + MI, ox = P.tx * TW + li;
    const bool inb = oy < a.H && ox < a.W;
    const unsigned pixb = inb ? (unsigned)((oy * a.W + ox) * a.y_cs) * 2u : OOB;       // byte offset of the pixel in the bf16 result
    const int cbase = co0 + wn * 32;                                                    // first cout of this wave's block
    unsigned hx[4], hy[4];          // packed bf16: hx[q] = couts (e 0, 1), hy[q] = (e 2, 3) of quad q
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      hx[q] = ph[MI][2 * q];
      hy[q] = ph[MI][2 * q + 1];
    }
    if (P64_DBG & 256) {            // timing ablation: nothing but the two result stores
#pragma unroll
      for (int pq = 0; pq < 2; ++pq) {
        const u32x4 o = {hx[2 * pq], hy[2 * pq], hx[2 * pq + 1], hy[2 * pq + 1]};
        const int co = cbase + 16 * pq + 8 * lh;
        __builtin_amdgcn_raw_buffer_store_b128(o, hrs, (co < a.Cout && pixb != OOB) ? pixb + (unsigned)co * 2u : OOB, 0, 0);
      }
      return;
    }
    if constexpr (MODE == 2) {      // one-bit ReLU mask: bit 8 q + 4 lh + e of the block's word -> 16-bit lanes of all ones / zeros
      const unsigned w2 = mw[MI] >> (4 * lh);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m0 = (int)(w2 << (31 - 8 * q)) >> 31, m1 = (int)(w2 << (30 - 8 * q)) >> 31;      // v_bfe_i32: 0 or -1
        const int m2 = (int)(w2 << (29 - 8 * q)) >> 31, m3 = (int)(w2 << (28 - 8 * q)) >> 31;
        hx[q] &= __builtin_amdgcn_perm((unsigned)m1, (unsigned)m0, 0x05040100u);
        hy[q] &= __builtin_amdgcn_perm((unsigned)m3, (unsigned)m2, 0x05040100u);
      }
    }
    if (a.relu) {                   // on the rounded pairs: max(int16 pattern, 0) -- negative values and -0 have the sign bit set (RNE and ReLU commute)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        hx[q] = pk_relu(hx[q], K0);
        hy[q] = pk_relu(hy[q], K0);
      }
    }
    if constexpr (MODE == 0) {      // sign bits of the stored values (post-ReLU: > 0 <=> pattern != 0)
      const int bw = a.y_cs >> 5;
      const size_t img_words = (size_t)a.H * a.W * bw;
      const __amdgpu_buffer_rsrc_t ybrs = __builtin_amdgcn_make_buffer_rsrc(a.y_bits != nullptr ? (void*)(a.y_bits + pn * img_words) : anyp, 0,
                                                                            (a.y_bits != nullptr && live) ? (int)(img_words * 4) : 0, 0x00020000);
      unsigned ybits = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned w = pk_min_u16(hx[q], K1) | (pk_min_u16(hy[q], K1) << 2);        // per half: 1 if the pattern is non-zero; bits 0 (e0), 16 (e1), 2 (e2), 18 (e3)
        ybits |= ((w | (w >> 15)) & 0xFu) << (8 * q);
      }
      const unsigned bitoff = (inb && cbase < a.Cout) ? (unsigned)(oy * a.W + ox) * (unsigned)(bw * 4) + (unsigned)(cbase >> 5) * 4u : OOB;
      mb_store(ybrs, bitoff, ybits, lh);
    }
    // lanes l and l + 32 hold the two halves of every 8-cout group: after the swap each lane owns 8 consecutive couts (16 pq + 8 lh ..) = one 16-byte store
#pragma unroll
    for (int pq = 0; pq < 2; ++pq) {
      const auto sx = __builtin_amdgcn_permlane32_swap(hx[2 * pq], hx[2 * pq + 1], false, false);
      const auto sy = __builtin_amdgcn_permlane32_swap(hy[2 * pq], hy[2 * pq + 1], false, false);
      const u32x4 o = {sx[0], sy[0], sx[1], sy[1]};
      const int co = cbase + 16 * pq + 8 * lh;
      __builtin_amdgcn_raw_buffer_store_b128(o, hrs, (co < a.Cout && pixb != OOB) ? pixb + (unsigned)co * 2u : OOB, 0, 0);
      if constexpr (MODE == 1) keep[MI & 1][pq] = inb ? o : u32x4{0, 0, 0, 0};
    }
    if constexpr (MODE == 1 && (MI & 1) == 1) {
      // fused forward pool of rows (MI - 1, MI): post-ReLU values are >= 0, positions outside the image count as 0, u16 max = bf16 max
      const size_t poimg = (size_t)PHo * PWo * a.y_cs;
      const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.pooled + pn * poimg), 0, live ? (int)(poimg * 2) : 0, 0x00020000);
      const __amdgpu_buffer_rsrc_t pcrs = __builtin_amdgcn_make_buffer_rsrc(pool_code ? (void*)(a.pool_code + pn * poimg) : anyp, 0, (pool_code && live) ? (int)poimg : 0, 0x00020000);
      const int ty = oy - 1;                                                            // top row of the window pair (even)
      const bool writer = (li & 1) == 0 && ty < a.H && ox < a.W;
      const unsigned ppix = writer ? (unsigned)(((ty >> 1) * PWo + (ox >> 1)) * a.y_cs) : OOB;     // element offset of the pooled pixel
#pragma unroll
      for (int pq = 0; pq < 2; ++pq) {
        const u32x4 r0 = keep[0][pq], r1 = keep[1][pq];
        u32x4 n0, n1, m;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          n0[k] = dpp_xor1(r0[k]);
          n1[k] = dpp_xor1(r1[k]);
          m[k] = pk_max_u16(pk_max_u16(r0[k], r1[k]), pk_max_u16(n0[k], n1[k]));
        }
        const int co = cbase + 16 * pq + 8 * lh;
        const bool ok = co < a.Cout && ppix != OOB;
        // code bytes (pool.hip): bits 1:0 = first maximum in scan order (0,0) (0,1) (1,0) (1,1), bits 5:2 = "input at that position > 0".
        // The writer lane is the LEFT column: a = r0, b = n0, c = r1, d = n1.  first = ne_a (1 + ne_b (1 + ne_c)), ne_x = (x != max).
        unsigned cw[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned ne_a = pk_min_u16(pk_sub(m[k], r0[k]), K1), ne_b = pk_min_u16(pk_sub(m[k], n0[k]), K1), ne_c = pk_min_u16(pk_sub(m[k], r1[k]), K1);
          const unsigned first = pk_mad(ne_a, pk_mad(ne_b, ne_c, ne_b), ne_a);
          const unsigned z = pk_mad(pk_mad(pk_min_u16(n1[k], K1), K2, pk_min_u16(r1[k], K1)), K4, pk_mad(pk_min_u16(n0[k], K1), K2, pk_min_u16(r0[k], K1)));
          cw[k] = pk_mad(z, K4, first);                                                 // two code values, one per half
        }
        // low bytes of the four halves pairs -> two dwords of four code bytes each
        const unsigned c0 = __builtin_amdgcn_perm(cw[1], cw[0], 0x06040200u), c1 = __builtin_amdgcn_perm(cw[3], cw[2], 0x06040200u);
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{c0, c1}, pcrs, ok ? ppix + (unsigned)co : OOB, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(m, prs, ok ? (ppix + (unsigned)co) * 2u : OOB, 0, 0);
      }
    }
  };

  // the tile just finished becomes the pending one: accumulators + bias -> pend, (MODE 2) its four mask words requested
  auto retire_tile = [&](const Tile& T) __attribute__((always_inline)) {
    if (P64_DBG & 64) {
      P = T;
      have_pend = true;
      return;
    }
    f32x4 bv[4];                    // couts 8 q + 4 lh + e of block wn
#pragma unroll
    for (int q = 0; q < 4; ++q) bv[q] = __builtin_bit_cast(f32x4, reinterpret_cast<const u32x4*>(smem)[BIAS_BASE + wn * 8 + 2 * q + lh]);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        bf16x2_t p0, p1;
        p0[0] = (__bf16)(acc[mi][4 * q] + bv[q][0]); p0[1] = (__bf16)(acc[mi][4 * q + 1] + bv[q][1]);
        p1[0] = (__bf16)(acc[mi][4 * q + 2] + bv[q][2]); p1[1] = (__bf16)(acc[mi][4 * q + 3] + bv[q][3]);
        ph[mi][2 * q] = __builtin_bit_cast(unsigned, p0);
        ph[mi][2 * q + 1] = __builtin_bit_cast(unsigned, p1);
      }
    P = T;
    have_pend = true;
    if constexpr (MODE == 2) {
      const int bw = a.y_cs >> 5;
      const size_t img_words = (size_t)a.H * a.W * bw;
      const bool tlive = T.n < a.N;
      const __amdgpu_buffer_rsrc_t mbrs = __builtin_amdgcn_make_buffer_rsrc((void*)const_cast<unsigned*>(a.mask_bits + (tlive ? T.n : 0) * img_words), 0,
                                                                            tlive ? (int)(img_words * 4) : 0, 0x00020000);
      const int cbase = co0 + wn * 32;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int oy = T.ty * TH + wm * 4 + mi, ox = T.tx * TW + li;
        const unsigned bitoff = (oy < a.H && ox < a.W && cbase < a.Cout) ? (unsigned)(oy * a.W + ox) * (unsigned)(bw * 4) + (unsigned)(cbase >> 5) * 4u : OOB;
        mw[mi] = mb_load(mbrs, bitoff);
      }
    }
  };

#ifdef P64_PROF
  unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tq, tp = __builtin_amdgcn_s_memtime();
  const unsigned long long t_begin = tp;
#define PMARK(k) do { tq = __builtin_amdgcn_s_memtime(); pt[k] += tq - tp; tp = tq; } while (0)
#else
#define PMARK(k) do { } while (0)
#endif
  Tile mcur = first;                            // multiply-side tile cursor
  for (int ti = 0; ti < my_tiles; ++ti) {
    const int phase = ti == 0 ? 0 : (ti == 1 ? 1 : 2);
    auto iteration = [&](auto KC_) __attribute__((always_inline)) {
      constexpr int KC = decltype(KC_)::value;
      if (grp == 0) {
        mfma_chunk(KC_);
        PMARK(0);
        if (have_pend && !(P64_DBG & 16)) piece(KC_);
        PMARK(1);
        if (phase == 0) wait_vm<allowed_vm(MODE, 0, KC, 0)>();
        else if (phase == 1) wait_vm<allowed_vm(MODE, 0, KC, 1)>();
        else wait_vm<allowed_vm(MODE, 0, KC, 2)>();
      } else {
        if (have_pend && !(P64_DBG & 16)) piece(KC_);
        PMARK(1);
        mfma_chunk(KC_);
        PMARK(0);
        if (phase == 0) wait_vm<allowed_vm(MODE, 1, KC, 0)>();
        else if (phase == 1) wait_vm<allowed_vm(MODE, 1, KC, 1)>();
        else wait_vm<allowed_vm(MODE, 1, KC, 2)>();
      }
      PMARK(2);
      issue_advance();              // (scalar bookkeeping only: the next tile's source offsets when the stream crosses a tile boundary)
      PMARK(6);
      if (!(P64_DBG & 1024)) __syncthreads();              // stream chunk (this + 1) has landed for everybody, and everybody is done with this chunk's buffer
      PMARK(3);
    };
    iteration(std::integral_constant<int, 0>{});
    iteration(std::integral_constant<int, 1>{});
    iteration(std::integral_constant<int, 2>{});
    iteration(std::integral_constant<int, 3>{});
    retire_tile(mcur);
    step(mcur);
    PMARK(4);
  }
  if (have_pend && !(P64_DBG & 16)) {      // the last tile's epilogue
    piece(std::integral_constant<int, 0>{});
    piece(std::integral_constant<int, 1>{});
    piece(std::integral_constant<int, 2>{});
    piece(std::integral_constant<int, 3>{});
  }
#ifdef P64_PROF
  PMARK(5);
  if (a.prof != nullptr && lane == 0) {
    unsigned long long* q = a.prof + ((size_t)blockIdx.x * NW + wave) * 10;
    for (int k = 0; k < 7; ++k) q[k] = pt[k];
    q[7] = tp - t_begin;
    q[8] = (unsigned long long)my_tiles;
    q[9] = (unsigned long long)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);      // HW_ID.SIMD_ID of this wave
  }
#endif
}

template <int MODE>
int launch(const P64Args& a0, hipStream_t stream) {
  static bool attr_set_dev[OSVOS_MAX_DEVICES] = {};
  bool& attr_set = attr_set_dev[osvos_current_device()];
  if (!attr_set) {
    OSVOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_bf16_p64_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    attr_set = true;
  }
  P64Args a = a0;
  a.tiles_x = ceil_div(a.W, TW);
  a.tiles_y = ceil_div(a.H, TH);
  a.nct = ceil_div(a.CoutP, BN);
  a.nsp = a.tiles_x * a.tiles_y * a.N;
  a.band = (a.nsp + 7) / 8;
  const long blocks = a.map == 0 ? (long)a.nct * a.nsp : (long)a.nct * a.band * 8;
  OSVOS_ARG_CHECK(blocks > 0 && blocks < (1L << 31), "conv3x3 bf16 p64: grid of %ld blocks", blocks);
  a.ntiles = (int)blocks;
  // one workgroup per CU; a workgroup keeps ONE cout tile (its filter is resident): map 0 needs G % nct == 0, map 1 (G / 8) % nct == 0
  const int n_cu = osvos_cu_count();
  const int gmul = 8 * a.nct;
  const long gmax = (long)n_cu / gmul * gmul;
  OSVOS_ARG_CHECK(gmax > 0, "conv3x3 bf16 p64: %d cout tiles do not fit a persistent grid of %d workgroups", a.nct, n_cu);
  const long grid = blocks > gmax ? gmax : blocks;
  hipLaunchKernelGGL((conv3x3_bf16_p64_kernel<MODE>), dim3((unsigned)grid), dim3(NT), LDS_BYTES, stream, a);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

}  // namespace

#ifdef P64_PROF
static unsigned long long* g_p64_prof = nullptr;
extern "C" void osvos_debug_set_p64_prof(void* p) { g_p64_prof = (unsigned long long*)p; }
#define P64_PROF_PTR g_p64_prof
#else
#define P64_PROF_PTR nullptr
#endif

// what the kernel takes: Cin = 64, bf16 in / out only (no fp32 result, no full-tensor mask), dense 8-channel-aligned result, sign bits only with ReLU
bool osvos_conv3x3_bf16_p64_applicable(int Cin, int Cout, int y_cs, bool has_y_f32, bool has_tensor_mask, bool has_mask_bits, bool has_y_bits, bool has_pool,
                                       int relu) {
  if (Cin != 64 || Cout % 8 != 0 || y_cs % 8 != 0 || has_y_f32 || has_tensor_mask) return false;
  if ((has_mask_bits || has_y_bits) && (Cout % 32 != 0 || y_cs != Cout)) return false;
  if (has_y_bits && !relu) return false;
  if (has_pool && (!relu || has_mask_bits || has_y_bits || y_cs != Cout)) return false;
  if (has_mask_bits && has_y_bits) return false;
  return true;
}

int osvos_conv3x3_bf16_p64(const void* x, const void* wpk, const float* bias, const unsigned* mask_bits, void* ybf, unsigned* y_bits, void* pooled_bf16,
                           void* pool_code, int N, int H, int W, int Cout, int y_cs, int relu, int map, hipStream_t stream) {
  OSVOS_ARG_CHECK(x && wpk && ybf, "conv3x3 bf16 p64: null pointer");
  OSVOS_ARG_CHECK(N > 0 && H > 0 && W > 0 && y_cs >= Cout, "conv3x3 bf16 p64: bad shape");
  OSVOS_ARG_CHECK(osvos_conv3x3_bf16_p64_applicable(64, Cout, y_cs, false, false, mask_bits != nullptr, y_bits != nullptr, pooled_bf16 != nullptr, relu),
                  "conv3x3 bf16 p64: unsupported epilogue combination (Cout %d, stride %d, relu %d)", Cout, y_cs, relu);
  OSVOS_ARG_CHECK(pool_code == nullptr || pooled_bf16 != nullptr, "conv3x3 bf16 p64: pool code bytes without a pooled result");
  OSVOS_ARG_CHECK((long)H * W * 64 < (1L << 29) && (long)H * W * y_cs < (1L << 29), "conv3x3 bf16 p64: image too large for 31-bit byte offsets");
  P64Args a;
  a.x = reinterpret_cast<const bf16_t*>(x); a.wpk = reinterpret_cast<const uint4*>(wpk); a.bias = bias; a.ybf = reinterpret_cast<bf16_t*>(ybf);
  a.N = N; a.H = H; a.W = W; a.Cout = Cout; a.CoutP = osvos_cout_pad(Cout); a.y_cs = y_cs;
  OSVOS_ENV_INT(band, "OSVOS_P64_BAND", 1);      // 0: XCD-local requests keep the interleaved map 1
  a.relu = relu; a.map = map ? (band ? 2 : 1) : 0;
  a.mask_bits = mask_bits; a.y_bits = y_bits; a.pooled = reinterpret_cast<bf16_t*>(pooled_bf16); a.pool_code = reinterpret_cast<unsigned char*>(pool_code);
  a.prof = P64_PROF_PTR;
  if (pooled_bf16 != nullptr) return launch<1>(a, stream);
  if (mask_bits != nullptr) return launch<2>(a, stream);
  return launch<0>(a, stream);
}
