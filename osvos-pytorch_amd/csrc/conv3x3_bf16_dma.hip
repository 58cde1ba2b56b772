// 3x3 convolution on bf16 MFMA operands, bf16 activations in HBM, operands staged by LDS-DMA.
//
// Same implicit GEMM as conv3x3_bf16.hip (halo tile of the input in LDS as planes of 16-byte channel groups, a tap =
// an LDS address offset, weights packed [tap][Cin/8][CoutP][8], cout-major accumulators, 16-byte buffer-store epilogue)
// but the global -> LDS path never touches a register: every wave issues `buffer_load_dwordx4 ... lds` (lane l of an
// instruction lands in LDS slot base + l; out-of-range offsets land as zeros -- tests/test_gpu_ops.py::
// test_lds_dma_layout_probe), three LDS buffers rotate (the DMA of chunk k+2 is in flight while chunk k is multiplied),
// and there is ONE barrier per 16-channel K chunk.  The register-staged kernel spends 20-40 % of a wave's time issuing
// loads, converting and writing LDS, two barriers per chunk (profiles/r01_phase_probe_bf16.txt); with its loads removed
// it runs 1435 TFLOP/s against 883 with them.  Here the wave's instruction stream is MFMA + ds_read + 12 DMA issues.
//
// Workgroup: 4 waves, 256 pixels (8 rows x 32) x NB*32 couts; wave (wm, wn) owns 4 rows x NB/2 cout blocks.
// LDS: 3 x 48 KB (NB = 4) -> one workgroup per CU, one wave per SIMD; latency is hidden by the DMA distance (two chunks)
// and by software-pipelined fragment reads.  Needs Cin % 16 == 0 (every layer but conv1_1).
//
// RESIDENT-FILTER form (CINR = 64, round 6): the Cin = 64 layers (conv1_2, conv2_1 and conv1_2's data gradient) are HBM-side
// (288 FLOP per byte at batch 12) and K = 576 is only four chunks, so a workgroup's life was mostly prologue + epilogue and every one of the
// 19,215 small tiles re-streamed the whole 73.7 KB filter of its 64 couts from L2 (1.4 GB of L2 -> LDS traffic per launch, as much as the
// tensor traffic itself).  Here the workgroups are PERSISTENT (one per CU, tiles b, b + G, ... all of ONE cout tile), the filter is loaded ONCE
// into its own LDS region (9 x Cin/8 x 64 slots = 73.7 KB), only the 16-channel activation chunks rotate (3 buffers of 21.5 KB, DMA two chunks
// ahead, crossing tile boundaries), and the next tile's first chunks are in flight while the epilogue stores drain.
#include "common.h"
#include "maskbits.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct DmaArgs {
  const bf16_t* x;       // bf16 NHWC, channel stride Cin (multiple of 16)
  const uint4* wpk;      // bf16 pack [9][CinP/8][CoutP][8]
  const float* bias;
  const void* mask;      // NHWC like y: fp32, or bf16 when mask_bf16
  float* y;              // may be NULL when ybf is given
  bf16_t* ybf;
  int N, H, W, Cin, CinP, Cout, CoutP, y_cs;
  int tiles_x, tiles_y, nct, nsp, map, ntiles;
  int relu, mask_bf16;
  const unsigned* mask_bits;   // one-bit-per-element ReLU mask (maskbits.h); takes precedence over `mask`
  unsigned* y_bits;            // optional: sign bits of the result
  bf16_t* pooled;              // optional: maxpool2x2 (ceil mode) of the bf16 result, [N][ceil(H/2)][ceil(W/2)][y_cs] (last convolution of a stage; needs ReLU)
  unsigned char* pool_code;    // optional, with pooled: one byte per pooled element for the pool's backward (pool.hip: first-max position + 4 "input > 0" bits)
};

constexpr int TW = 32, HWD = TW + 2;
constexpr int KG = 2;                                                               // 16 channels per chunk
constexpr unsigned OOB = 0x80000000u;

// WGM = 2: 4 waves, 8 rows x 32 px, three rotating LDS buffers, one wave per SIMD
// WGM = 4: 8 waves, 16 rows x 32 px, two LDS buffers, two waves per SIMD (each covers the other's DMA issue and LDS waits)
// CINR = 0: the weights of a chunk travel with its activations (B_SLOTS per rotating buffer); CINR = Cin > 0: the whole filter is resident
template <int NB, int WGM, int CINR = 0>
struct DmaCfg {
  static constexpr bool RES = CINR > 0;
  static constexpr int NW = 2 * WGM, NT = 64 * NW;         // wave grid WGM x 2
  static constexpr int TH = 4 * WGM, HHT = TH + 2, PLANE = HHT * HWD;
  static constexpr int A_SLOTS = (KG * PLANE + 63) / 64 * 64;
  static constexpr int A_INSTR = A_SLOTS / 64;
  static constexpr int NBUF = (RES || WGM == 2) ? 3 : 2;
  static constexpr int BN = NB * 32;
  static constexpr int WN = NB / 2;                        // cout blocks per wave
  static constexpr int WM = 4;                             // rows per wave
  static constexpr int B_SLOTS = 9 * KG * BN;              // multiple of 64 (one chunk's weights)
  static constexpr int B_INSTR = B_SLOTS / 64;
  static constexpr int BUF_B = RES ? 0 : B_SLOTS;          // weight slots inside a rotating buffer
  static constexpr int BUF_SLOTS = A_SLOTS + BUF_B + 64;   // + one spare instruction target (keeps every wave's DMA count equal)
  static constexpr int NA = (A_INSTR + NW - 1) / NW, NBI = RES ? 0 : (B_INSTR + NW - 1) / NW;
  static constexpr int NDMA = NA + NBI;                    // DMA instructions per wave per chunk
  static constexpr int RCG = CINR / 8;                     // resident filter: 8-channel groups of the reduction dimension
  static constexpr int RB_BASE = NBUF * BUF_SLOTS, RB_SLOTS = 9 * RCG * BN, RB_INSTR = RB_SLOTS / 64;
  static constexpr size_t LDS_BYTES = (size_t)(NBUF * BUF_SLOTS + RB_SLOTS) * 16;
  static_assert(CINR % 16 == 0, "whole chunks");
  static_assert(B_SLOTS % 64 == 0, "weight tile must be whole DMA instructions");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <int NB, int WGM, int CINR>
__global__ __launch_bounds__(64 * 2 * WGM, WGM == 4 ? 2 : 1) void conv3x3_bf16_dma_kernel(DmaArgs a) {
  using C = DmaCfg<NB, WGM, CINR>;
  constexpr bool RES = C::RES;
  constexpr int TH = C::TH, PLANE = C::PLANE, A_SLOTS = C::A_SLOTS, A_INSTR = C::A_INSTR, NBUF = C::NBUF, NW = C::NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint4* lds = reinterpret_cast<const uint4*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;

  // Tile index -> (image, tile row, tile column, cout tile).  map 0: cout tile in the low bits; map 1: eight consecutive indices
  // are eight spatial tiles (one per XCD: block b runs on XCD b % 8) and the cout tiles of a spatial tile follow on the SAME XCD.
  // A persistent workgroup walks t = blockIdx.x, blockIdx.x + gridDim.x, ... (gridDim.x a multiple of 8 keeps it on its XCD).
  struct Tile { int n, x0, y0, co0; bool live; };
  auto decode = [&](int t) -> Tile {
    int sp, ct;
    if (a.map == 0) {
      sp = t / a.nct;
      ct = t % a.nct;
    } else {
      const int j = t >> 3;
      ct = j % a.nct;
      sp = (j / a.nct) * 8 + (t & 7);
    }
    Tile r;
    r.live = sp < a.nsp;
    const int tx = sp % a.tiles_x;
    sp /= a.tiles_x;
    r.x0 = tx * TW;
    r.y0 = (sp % a.tiles_y) * TH;
    r.n = sp / a.tiles_y;
    r.co0 = ct * C::BN;
    return r;
  };
  const int CG = a.CinP >> 3;
  const int ntiles = a.ntiles;
  int my_tiles = 0;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) ++my_tiles;      // (scalar loop; a handful of iterations)

  // The DMA instructions are issued through inline asm: hipcc tracks `__builtin_amdgcn_raw_ptr_buffer_load_lds` as an LDS store
  // and puts `s_waitcnt vmcnt(0)` in front of the next ds_read -- which would wait for the chunk that was just requested and
  // serialise the pipeline.  Hidden from the compiler, the ordering is ours to keep: vmcnt(NDMA) + barrier per chunk (below).
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  auto make_rsrc = [](const void* p, int bytes) -> i32x4 {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    return i32x4{(int)(unsigned)v, (int)(unsigned)(v >> 32), bytes, 0x00020000};
  };
  const size_t ximg_elems = (size_t)a.H * a.W * a.Cin;
  const i32x4 wrs = make_rsrc(a.wpk, (int)((size_t)9 * CG * a.CoutP * 16));
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;     // LDS byte address of the dynamic segment
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  auto dma16 = [](const i32x4& rs, unsigned lds_addr, unsigned voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "m0", "memory");
  };
#pragma clang diagnostic pop

  // per-lane source offsets of this wave's DMA instructions for the tile being ISSUED (chunk 0; the chunk advance rides in the
  // scalar offset).  The issue side runs DIST chunks ahead of the multiply side and crosses tile boundaries on its own.
  unsigned a_off[C::NA], b_off[C::NBI > 0 ? C::NBI : 1];
  i32x4 xrs = make_rsrc(a.x, 0);
  auto set_issue_tile = [&](const Tile& T) {
    xrs = make_rsrc(a.x + (size_t)T.n * ximg_elems, (int)(ximg_elems * 2));
#pragma unroll
    for (int i = 0; i < C::NA; ++i) {
      const int e = 64 * (wave + NW * i) + lane;             // slot inside the A region (instruction wave + NW i)
      const int g = e / PLANE, rem = e % PLANE;
      const int hy = rem / HWD, hx = rem % HWD;
      const int gy = T.y0 + hy - 1, gx = T.x0 + hx - 1;
      a_off[i] = (T.live && e < KG * PLANE && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? (unsigned)(((gy * a.W + gx) * a.Cin + 8 * g) * 2) : OOB;
    }
#pragma unroll
    for (int i = 0; i < C::NBI; ++i) {
      const int e = 64 * (wave + NW * i) + lane;
      const int tap = e / (KG * C::BN), rem = e % (KG * C::BN);
      const int g = rem / C::BN, nn = rem % C::BN;
      b_off[i] = (T.live && e < C::B_SLOTS && T.co0 + nn < a.CoutP) ? (unsigned)(((tap * CG + g) * a.CoutP + T.co0 + nn) * 16) : OOB;
    }
  };
  // LDS target of instruction i of this wave inside a buffer (wave-uniform); instructions past the region go to the spare slots
  const int wv = __builtin_amdgcn_readfirstlane(wave);        // wave-uniform by construction; make it a scalar for m0
  // the d-th DMA instruction of this wave for chunk kc into buffer buf (d is a compile-time constant after unrolling)
  auto dma_one = [&](int d, int kc, int buf, unsigned dead) {
    const unsigned base = lds0 + (unsigned)(buf * C::BUF_SLOTS * 16);
    if (d < C::NA) {
      const int j = wv + NW * d;
      dma16(xrs, base + (unsigned)(j < A_INSTR ? j * 1024 : (A_SLOTS + C::BUF_B) * 16), a_off[d] | dead, kc * (8 * KG * 2));
    } else if constexpr (!RES) {
      const int i = d - C::NA, j = wv + NW * i;
      dma16(wrs, base + (unsigned)(j < C::B_INSTR ? (A_SLOTS + 64 * j) * 16 : (A_SLOTS + C::BUF_B) * 16), b_off[i] | dead, kc * KG * a.CoutP * 16);
    }
  };
  auto dma_chunk = [&](int kc, int buf, unsigned dead) {
#pragma unroll
    for (int d = 0; d < C::NDMA; ++d) dma_one(d, kc, buf, dead);
  };

  const int a_idx = lh * PLANE + (wm * C::WM) * HWD + li;              // + mi * HWD + r * HWD + s
  const int b_idx = A_SLOTS + lh * C::BN + wn * C::WN * 32 + li;       // + tap * KG * BN + ni * 32
  const int rb_idx = C::RB_BASE + lh * C::BN + wn * C::WN * 32 + li;   // resident filter: + (tap * RCG + kc * KG) * BN + ni * 32

  f32x16 acc[C::WM][C::WN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int mi = 0; mi < C::WM; ++mi)
#pragma unroll
      for (int ni = 0; ni < C::WN; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
  };
  zero_acc();

  const int nchunks = a.Cin >> 4;
  const int total = my_tiles * nchunks;                       // the chunk stream of this workgroup, across its tiles
  // DIST = how many chunks the DMA runs ahead (NBUF - 1).  s_waitcnt vmcnt((DIST - 1) * NDMA): everything but the newest
  // DIST - 1 chunks' DMA instructions of THIS wave has landed (epilogue loads / stores in between only make the wait stricter)
  constexpr int DIST = NBUF - 1;
  constexpr int VM = (DIST - 1) * C::NDMA;
  constexpr int WAITCNT = 0x0F70 | (VM & 15) | ((VM >> 4) << 14);
  // issue side: chunk stream position gi = (tile it_, chunk ikc)
  int it_ = blockIdx.x, ikc = 0, gi = 0;
  if constexpr (RES) {
    // the whole filter of THIS workgroup's cout tile (every tile it walks has the same one: the launcher keeps gridDim.x a multiple of 8 nct), once.
    // Issued ahead of the first activation chunks: the first `s_waitcnt vmcnt` below covers it (VM counts complete in order).
    const int co0 = decode(blockIdx.x).co0;
    for (int j = wv; j < C::RB_INSTR; j += NW) {
      const int e = 64 * j + lane;
      const int tap = e / (C::RCG * C::BN), rem = e % (C::RCG * C::BN);
      const int g = rem / C::BN, nn = rem % C::BN;
      dma16(wrs, lds0 + (unsigned)((C::RB_BASE + 64 * j) * 16), co0 + nn < a.CoutP ? (unsigned)(((tap * CG + g) * a.CoutP + co0 + nn) * 16) : OOB, 0);
    }
  }
  set_issue_tile(decode(it_));
  auto issue_advance = [&]() {                               // after all DMA instructions of stream chunk gi have been issued
    ++gi;
    if (++ikc == nchunks) {
      ikc = 0;
      it_ += gridDim.x;
      if (gi < total) set_issue_tile(decode(it_));
    }
  };
#pragma unroll
  for (int c = 0; c < DIST; ++c) {
    dma_chunk(ikc, c, gi < total ? 0u : OOB);
    issue_advance();
  }
  __builtin_amdgcn_s_waitcnt(WAITCNT);
  __syncthreads();
  int cur = 0;                                                 // stream chunk g lives in buffer g % NBUF
  int tcur = blockIdx.x, kc = 0;                               // multiply side: tile, chunk
  for (int g = 0; g < total; ++g) {
    // stream chunk g + DIST goes to the buffer that was read in iteration g - 1; every wave left that iteration through the barrier
    // below.  Its DMA instructions are issued one at a time between the MFMAs: back to back they keep the wave off the matrix pipe.
    const int tgt = cur == 0 ? NBUF - 1 : cur - 1;
    const unsigned dead = gi < total ? 0u : OOB;
    const int dkc = ikc;
    const uint4* As = lds + (size_t)cur * C::BUF_SLOTS;
    // 9 tap stages, software pipelined: the fragments of tap t+1 are requested before the MFMAs of tap t issue
    uint4 fa[2][C::WM], fb[2][C::WN];
    auto ldfrag = [&](int tap, int set) {
      const int r = tap / 3, s = tap % 3;
#pragma unroll
      for (int mi = 0; mi < C::WM; ++mi) fa[set][mi] = As[a_idx + (mi + r) * HWD + s];
#pragma unroll
      for (int ni = 0; ni < C::WN; ++ni)
        fb[set][ni] = RES ? lds[rb_idx + (tap * C::RCG + kc * KG) * C::BN + ni * 32] : As[b_idx + tap * KG * C::BN + ni * 32];
    };
    ldfrag(0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap + 1 < 9) ldfrag(tap + 1, (tap + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      constexpr int PER = C::WM * C::WN;                       // MFMAs per tap
      constexpr int GAP = (9 * PER) / C::NDMA;                 // one DMA issue every GAP MFMAs
#pragma unroll
      for (int mi = 0; mi < C::WM; ++mi)
#pragma unroll
        for (int ni = 0; ni < C::WN; ++ni) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb[tap & 1][ni]),
                                                                __builtin_bit_cast(bf16x8_t, fa[tap & 1][mi]), acc[mi][ni], 0, 0, 0);
          const int m = tap * PER + mi * C::WN + ni;           // index of the MFMA just issued
          if (m % GAP == GAP / 2 && m / GAP < C::NDMA) {
            dma_one(m / GAP, dkc, tgt, dead);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    issue_advance();
    __builtin_amdgcn_s_waitcnt(WAITCNT);                       // stream chunk g+1 has landed (this wave's part) ...
    __syncthreads();                                           // ... everybody's part has, and everybody is done with `cur`
    cur = cur == NBUF - 1 ? 0 : cur + 1;
    if (++kc < nchunks) continue;
    kc = 0;
    // ---- this tile is complete: epilogue (the DMA of the next tile's first chunks is already in flight) ----
    const Tile T = decode(tcur);
    tcur += gridDim.x;
    const int n = T.n, x0 = T.x0, y0 = T.y0, co0 = T.co0;
    if (T.live) {
    // ---- epilogue: cout-major accumulators (weights are the first MFMA operand), 16-byte raw buffer stores, no branches ----
    const size_t img_elems = (size_t)a.H * a.W * a.y_cs;
    void* const anyp = const_cast<uint4*>(a.wpk);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(a.y != nullptr ? (void*)(a.y + n * img_elems) : anyp, 0,
                                                                         a.y != nullptr ? (int)(img_elems * 4) : 0, 0x00020000);
    const int msz = a.mask_bf16 ? 2 : 4;
    const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(
        a.mask != nullptr ? (void*)(reinterpret_cast<char*>(const_cast<void*>(a.mask)) + n * img_elems * msz) : anyp, 0,
        a.mask != nullptr ? (int)(img_elems * msz) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(a.bias != nullptr ? (void*)const_cast<float*>(a.bias) : anyp, 0,
                                                                         a.bias != nullptr ? a.Cout * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(a.ybf != nullptr ? (void*)(a.ybf + n * img_elems) : anyp, 0,
                                                                         a.ybf != nullptr ? (int)(img_elems * 2) : 0, 0x00020000);
    const bool pool_fwd = a.pooled != nullptr;
    const int PHo = (a.H + 1) / 2, PWo = (a.W + 1) / 2;
    const size_t poimg = (size_t)PHo * PWo * a.y_cs;
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(pool_fwd ? (void*)(a.pooled + n * poimg) : anyp, 0, pool_fwd ? (int)(poimg * 2) : 0, 0x00020000);
    const bool pool_code = pool_fwd && a.pool_code != nullptr;
    const __amdgpu_buffer_rsrc_t pcrs = __builtin_amdgcn_make_buffer_rsrc(pool_code ? (void*)(a.pool_code + n * poimg) : anyp, 0, pool_code ? (int)poimg : 0, 0x00020000);
    // one-bit masks (maskbits.h): words per pixel = y_cs / 32
    const int bw = a.y_cs >> 5;
    const size_t img_words = (size_t)a.H * a.W * bw;
    const __amdgpu_buffer_rsrc_t mbrs = __builtin_amdgcn_make_buffer_rsrc(a.mask_bits != nullptr ? (void*)const_cast<unsigned*>(a.mask_bits + n * img_words) : anyp, 0,
                                                                          a.mask_bits != nullptr ? (int)(img_words * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t ybrs = __builtin_amdgcn_make_buffer_rsrc(a.y_bits != nullptr ? (void*)(a.y_bits + n * img_words) : anyp, 0,
                                                                          a.y_bits != nullptr ? (int)(img_words * 4) : 0, 0x00020000);
#pragma unroll
    for (int ni = 0; ni < C::WN; ++ni) {
      const int cb = co0 + (wn * C::WN + ni) * 32 + 4 * lh;
      f32x4 bv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) bv[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, (cb + 8 * q) * 4, 0, 0));
      u32x4 keep[C::WM][2];      // (pool forward only: the packed bf16 results of this wave, zero outside the image)
#pragma unroll
      for (int mi = 0; mi < C::WM; ++mi) {
        const int oy = y0 + wm * C::WM + mi, ox = x0 + li;
        const unsigned pix = (oy < a.H && ox < a.W) ? (unsigned)((oy * a.W + ox) * a.y_cs) * 4u : OOB;
        uint2 hb[4];
        const unsigned bitoff = (pix != OOB && cb - 4 * lh < a.Cout) ? (unsigned)(oy * a.W + ox) * (unsigned)(bw * 4) + (unsigned)((cb - 4 * lh) >> 5) * 4u : OOB;
        const unsigned mword = mb_load(mbrs, bitoff);      // (an absent tensor's descriptor drops the access)
        unsigned ybits = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int co = cb + 8 * q;
          const unsigned off = co < a.Cout ? pix + (unsigned)co * 4u : OOB;
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = acc[mi][ni][4 * q + e] + bv[q][e];
            if (a.relu) v[e] = v[e] > 0.f ? v[e] : 0.f;
          }
          if (a.mask_bits != nullptr) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = mb_test(mword, q, lh, e) ? v[e] : 0.f;
          } else if (a.mask != nullptr) {
            if (a.mask_bf16) {
              typedef short s16x4 __attribute__((ext_vector_type(4)));
              const s16x4 m = __builtin_bit_cast(s16x4, __builtin_amdgcn_raw_buffer_load_b64(mrs, off >> 1, 0, 0));
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = m[e] > 0 ? v[e] : 0.f;
            } else {
              const f32x4 m = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(mrs, off, 0, 0));
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = m[e] > 0.f ? v[e] : 0.f;
            }
          }
          if (a.y != nullptr) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, off, 0, 0);
          bf16x4_t h;
          h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
          hb[q] = __builtin_bit_cast(uint2, h);
          ybits |= a.ybf != nullptr ? mb_bits_bf16(hb[q], q) : mb_bits_f32(v, q);      // (the stored value decides: bf16 when that is what is kept)
        }
        if (a.y_bits != nullptr) mb_store(ybrs, bitoff, ybits, lh);
        if (a.ybf != nullptr) {
#pragma unroll
          for (int pq = 0; pq < 2; ++pq) {
            const auto sx = __builtin_amdgcn_permlane32_swap(hb[2 * pq].x, hb[2 * pq + 1].x, false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(hb[2 * pq].y, hb[2 * pq + 1].y, false, false);
            const u32x4 o = {sx[0], sy[0], sx[1], sy[1]};
            const int co = co0 + (wn * C::WN + ni) * 32 + 16 * pq + 8 * lh;
            const unsigned off = co < a.Cout ? (pix >> 1) + (unsigned)co * 2u : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(o, hrs, off, 0, 0);
            if (pool_fwd) keep[mi][pq] = pix != OOB ? o : u32x4{0, 0, 0, 0};
          }
        }
      }
      if (pool_fwd) {
        // fused forward pool (this launch is the last convolution of a stage): max over the 2 x 2 window of the packed bf16 results.  Post-ReLU
        // values are >= 0, so positions outside the image count as 0 and 16-bit UNSIGNED integer max is the bf16 max.
        typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
        constexpr int NP = C::WM / 2;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
          const int oy = y0 + wm * C::WM + 2 * j, ox = x0 + li;      // top row of the window pair
          const bool writer = (li & 1) == 0 && oy < a.H && ox < a.W;
          const unsigned ppix = writer ? (unsigned)(((oy >> 1) * PWo + (ox >> 1)) * a.y_cs) * 2u : OOB;
#pragma unroll
          for (int pq = 0; pq < 2; ++pq) {
            // this lane's column: r0 = its upper row, r1 = its lower row (rows 2j, 2j+1 of this wave); lane ^ 1 the neighbouring column
            const u32x4 r0 = keep[2 * j][pq], r1 = keep[2 * j + 1][pq];
            u16x8 m = __builtin_elementwise_max(__builtin_bit_cast(u16x8, r0), __builtin_bit_cast(u16x8, r1));
            const int co = co0 + (wn * C::WN + ni) * 32 + 16 * pq + 8 * lh;
            if (!pool_code) {
              const u32x4 t = __builtin_bit_cast(u32x4, m);
              const u32x4 u = {(unsigned)__shfl_xor((int)t[0], 1, 64), (unsigned)__shfl_xor((int)t[1], 1, 64), (unsigned)__shfl_xor((int)t[2], 1, 64),
                               (unsigned)__shfl_xor((int)t[3], 1, 64)};
              m = __builtin_elementwise_max(m, __builtin_bit_cast(u16x8, u));
            } else {
              // the neighbouring column's two rows arrive separately: the writer needs all four window values for the code byte (as conv3x3_bf16.hip)
              const u32x4 n0 = {(unsigned)__shfl_xor((int)r0[0], 1, 64), (unsigned)__shfl_xor((int)r0[1], 1, 64), (unsigned)__shfl_xor((int)r0[2], 1, 64),
                                (unsigned)__shfl_xor((int)r0[3], 1, 64)};
              const u32x4 n1 = {(unsigned)__shfl_xor((int)r1[0], 1, 64), (unsigned)__shfl_xor((int)r1[1], 1, 64), (unsigned)__shfl_xor((int)r1[2], 1, 64),
                                (unsigned)__shfl_xor((int)r1[3], 1, 64)};
              m = __builtin_elementwise_max(m, __builtin_elementwise_max(__builtin_bit_cast(u16x8, n0), __builtin_bit_cast(u16x8, n1)));
              unsigned cw[2] = {0u, 0u};
#pragma unroll
              for (int e = 0; e < 8; ++e) {      // scan order (0,0) (0,1) (1,0) (1,1), strict >: the first maximum (pool.hip); post-ReLU bf16 compare as u16
                const int sh = 16 * (e & 1);
                const unsigned av = (r0[e >> 1] >> sh) & 0xffffu, bv2 = (n0[e >> 1] >> sh) & 0xffffu;
                const unsigned cv = (r1[e >> 1] >> sh) & 0xffffu, dv = (n1[e >> 1] >> sh) & 0xffffu;
                unsigned bi = 0u, best = av;
                if (bv2 > best) { best = bv2; bi = 1u; }
                if (cv > best) { best = cv; bi = 2u; }
                if (dv > best) { bi = 3u; }
                const unsigned byte = bi | (av ? 4u : 0u) | (bv2 ? 8u : 0u) | (cv ? 16u : 0u) | (dv ? 32u : 0u);
                cw[e >> 2] |= byte << (8 * (e & 3));
              }
              typedef unsigned u32x2s __attribute__((ext_vector_type(2)));
              __builtin_amdgcn_raw_buffer_store_b64(u32x2s{cw[0], cw[1]}, pcrs, (co < a.Cout && ppix != OOB) ? (ppix >> 1) + (unsigned)co : OOB, 0, 0);
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, m), prs, (co < a.Cout && ppix != OOB) ? ppix + (unsigned)co * 2u : OOB, 0, 0);
          }
        }
      }
    }

    }
    zero_acc();
  }
}

template <int NB, int WGM, int CINR = 0>
int launch(const DmaArgs& a0, int persist, hipStream_t stream) {
  using C = DmaCfg<NB, WGM, CINR>;
  OSVOS_ARG_CHECK(CINR == 0 || a0.Cin == CINR, "conv3x3 bf16 dma: the resident-filter form is built for Cin = %d (got %d)", CINR, a0.Cin);
  constexpr int TH = C::TH;
  static bool attr_set_dev[OSVOS_MAX_DEVICES] = {};      // hipFuncSetAttribute is per device
  bool& attr_set = attr_set_dev[osvos_current_device()];
  if (!attr_set) {
    OSVOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_bf16_dma_kernel<NB, WGM, CINR>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)C::LDS_BYTES));
    attr_set = true;
  }
  DmaArgs a = a0;
  a.tiles_x = ceil_div(a.W, TW);
  a.tiles_y = ceil_div(a.H, TH);
  a.nct = ceil_div(a.CoutP, C::BN);
  a.nsp = a.tiles_x * a.tiles_y * a.N;
  const long blocks = a.map == 0 ? (long)a.nct * a.nsp : (long)a.nct * ((a.nsp + 7) / 8) * 8;
  OSVOS_ARG_CHECK(blocks > 0 && blocks < (1L << 31), "conv3x3 bf16 dma: grid of %ld blocks", blocks);
  a.ntiles = (int)blocks;
  // persistent: one workgroup per CU (the LDS footprint allows no more) walks tiles b, b + G, ...; the DMA of a tile's first chunks
  // overlaps the previous tile's epilogue.  G is a multiple of 8 so that a workgroup's tiles stay on its XCD's L2.
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    OSVOS_HIP_CHECK(hipGetDevice(&dev));
    OSVOS_HIP_CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    n_cu = n_cu / 8 * 8;
    if (n_cu < 8) n_cu = 8;
  }
  // (resident filter: a workgroup keeps ONE cout tile -- map 0 needs G % nct == 0, map 1 (G / 8) % nct == 0: G a multiple of 8 nct serves both)
  const int gmul = 8 * (CINR > 0 ? a.nct : 1);
  const long gmax = (long)n_cu / gmul * gmul;
  OSVOS_ARG_CHECK(gmax > 0, "conv3x3 bf16 dma: %d cout tiles do not fit a persistent grid of %d workgroups", a.nct, n_cu);
  const long grid = persist && blocks > gmax ? gmax : blocks;
  hipLaunchKernelGGL((conv3x3_bf16_dma_kernel<NB, WGM, CINR>), dim3((unsigned)grid), dim3(C::NT), C::LDS_BYTES, stream, a);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

}  // namespace

bool osvos_conv3x3_bf16_dma_applicable(int Cin, int Cout, int y_cs) { return Cin % 16 == 0 && Cout % 8 == 0 && y_cs % 8 == 0; }

// variant 0: 256 px x 128 couts (4 waves), 1: 256 px x 64 couts (4 waves), 2: 512 px x 128 couts (8 waves), 3: 512 px x 64 couts (8 waves),
// 4 / 5: variants 0 / 2 as persistent workgroups (one per CU, tiles pipelined back to back);
// map = 1: XCD-local spatial block order
// 6 / 7: RESIDENT-FILTER persistent forms for Cin = 64 (512 px / 256 px x 64 couts): see the head of this file
int osvos_conv3x3_bf16_dma(const void* x, const void* wpk, const float* bias, const void* mask, int mask_bf16, const unsigned* mask_bits, float* y, void* ybf,
                           unsigned* y_bits, void* pooled_bf16, int N, int H, int W, int Cin, int Cout, int y_cs, int relu, int variant, int map, hipStream_t stream,
                           void* pool_code) {
  OSVOS_ARG_CHECK(x && wpk && (y || ybf), "conv3x3 bf16 dma: null pointer");
  OSVOS_ARG_CHECK(N > 0 && H > 0 && W > 0 && osvos_conv3x3_bf16_dma_applicable(Cin, Cout, y_cs) && y_cs >= Cout,
                  "conv3x3 bf16 dma: needs Cin %% 16 == 0, Cout %% 8 == 0, y_cs %% 8 == 0 (got %d, %d, %d)", Cin, Cout, y_cs);
  OSVOS_ARG_CHECK((long)H * W * Cin < (1L << 29) && (long)H * W * y_cs < (1L << 29), "conv3x3 bf16 dma: image too large for 31-bit byte offsets");
  DmaArgs a;
  a.x = reinterpret_cast<const bf16_t*>(x); a.wpk = reinterpret_cast<const uint4*>(wpk); a.bias = bias; a.mask = mask; a.mask_bf16 = mask_bf16 ? 1 : 0;
  a.y = y; a.ybf = reinterpret_cast<bf16_t*>(ybf);
  a.mask_bits = mask_bits; a.y_bits = y_bits; a.pooled = reinterpret_cast<bf16_t*>(pooled_bf16);
  a.pool_code = reinterpret_cast<unsigned char*>(pool_code);
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.CinP = (Cin + 31) / 32 * 32; a.Cout = Cout; a.CoutP = osvos_cout_pad(Cout); a.y_cs = y_cs;
  a.relu = relu; a.map = map ? 1 : 0;
  switch (variant) {
    case 0: return launch<4, 2>(a, 0, stream);
    case 1: return launch<2, 2>(a, 0, stream);
    case 2: return launch<4, 4>(a, 0, stream);
    case 3: return launch<2, 4>(a, 0, stream);
    case 4: return launch<4, 2>(a, 1, stream);      // persistent forms of 0 and 2
    case 5: return launch<4, 4>(a, 1, stream);
    case 6: return launch<2, 4, 64>(a, 1, stream);  // resident 64-channel filter, persistent
    case 7: return launch<2, 2, 64>(a, 1, stream);
    default: osvos_set_error("conv3x3 bf16 dma: unknown variant %d", variant); return -1;
  }
}
