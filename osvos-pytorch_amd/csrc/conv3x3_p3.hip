// 3x3 stride-1 pad-1 convolution of the f32x3 arithmetic (conv3x3_f32x3.hip: fp32-grade results from six bf16 MFMA products per fp32
// product) on PRE-SPLIT operands: the activation arrives as a P3 tensor (p3.h: its three bf16 piece planes, formed once by the producer's
// epilogue), the filter as the pre-split pack of osvos_pack_x3.  Replaces the forward of nn.Conv2d(k=3, p=1) + ReLU (reference
// vgg_osvos.py:41,142-143) and, with the rotated pack, the data-gradient half of its backward (autograd of train_online.py:141).
//
// Nothing is converted, subtracted or carried through a register on the way to the matrix pipe: every wave issues
// `buffer_load_dwordx4 ... lds` (lane l of an instruction lands in LDS slot base + l; out-of-range lanes land as zeros --
// tests/test_gpu_ops.py::test_lds_dma_layout_probe), the wave's instruction stream is MFMA + ds_read_b128 + a few DMA issues.
//
//   * workgroup = 256 pixels (32 x 8, or 16 x 16 on narrow maps) x NB * 32 couts; 16-channel K chunks (one MFMA k-step)
//   * A: the (TH+2) x (TW+2) halo tile of a chunk, [piece 3][group 2][rows][cols] 16-byte slots (8 bf16 channels of one pixel), staged
//     once per chunk and re-used by the 9 taps (a tap = an LDS address offset); two buffers, chunk k+1 lands while chunk k multiplies
//   * B: the chunk's weights do NOT fit next to that twice (3 pieces x 9 taps x 16 ci x BN co = 110 KB at BN = 128), so they move by TAP
//     ROW: a stage = (chunk, kernel row r) = [piece 3][tap column 3][group 2][BN] slots = 37 KB, two buffers; the DMA of stage s+1
//     (and a third of the next chunk's A tile) is issued between the MFMAs of stage s; one s_waitcnt vmcnt(0) + barrier per stage
//   * per (tap, M block) step: 3 A fragments (+ 3 WN B fragments per tap) -> 6 WN MFMAs, fragment reads one step ahead (sched_barrier)
//   * 16-pixel-wide tiles keep the halo tile DENSE (pitch 18): the second row of an M block is rotated by two columns in the lane map
//     (lane 16 + i holds column (i - 2) mod 16), which puts the four 16-lane groups of a ds_read_b128 on 16 distinct slots mod 16
//   * epilogue: cout-major accumulators; fp32 NHWC and / or P3 output (pieces formed here, 16-byte stores per plane), bias / ReLU /
//     ReLU-mask of the producer (fp32, or plane 0 of its P3 tensor) fused; split-K partial sums + p3 finalize kernel
#include "common.h"
#include "kernels.h"
#include "p3.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct P3Args {
  const bf16_t* x;       // P3 [N][3][H][W][Cin]
  const uint4* wpk3;     // [piece 3][tap 9][Cin/8][CoutP][8 bf16] (osvos_pack_x3)
  const float* bias;
  const void* mask;      // fp32 NHWC with channel stride mask_cs, or (mask_p3) a P3 tensor with channel stride mask_cs: plane 0 is read
  float* y;              // fp32 NHWC, channel stride y_cs (may be NULL)
  bf16_t* y3;            // P3 [N][3][H][W][y3_cs] (may be NULL)
  float* part;           // split-K partial sums
  int N, H, W, Cin, Cout, CoutP, y_cs, y3_cs, mask_p3, mask_cs;
  int tiles_x, tiles_y, nct, nsp, map, relu, ksplit;
};

constexpr int cdivp(int a, int b) { return (a + b - 1) / b; }
constexpr unsigned OOB = 0x80000000u;

// scheduling groups of one (tap, M block) step: each of its NM MFMAs followed by its share of the NDS fragment reads of the NEXT step
template <int NM, int NDS, int I = 0>
__device__ __forceinline__ void step_groups() {
  if constexpr (I < NM) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    constexpr int nds = (NDS * (I + 1)) / NM - (NDS * I) / NM;
    if constexpr (nds > 0) __builtin_amdgcn_sched_group_barrier(0x100, nds, 0);
    step_groups<NM, NDS, I + 1>();
  }
}

// ILV = 1: the next step's fragment reads are interleaved with the step's MFMAs (sched_group_barrier) instead of being issued as a
// block in front of them -- what a one-wave-per-SIMD workgroup needs (nothing else covers the issue time of that block)
template <int RBW_, int NB_, int WGM_, int WGN_, int ILV_ = 0>
struct PCfg {
  static constexpr int RBW = RBW_, NB = NB_, WGM = WGM_, WGN = WGN_, ILV = ILV_;
  static constexpr int NW = WGM * WGN, NT = 64 * NW;
  static constexpr int RBH = 32 / RBW;                      // rows of one 32-pixel M block
  static constexpr int TW = RBW, TH = 8 * RBH;              // eight M blocks stacked vertically: 32 x 8 or 16 x 16 pixels
  static constexpr int HWD = TW + 2, HHT = TH + 2, PLANE = HHT * HWD;
  static constexpr int A_USED = 6 * PLANE;                  // [piece 3][group 2][PLANE]
  static constexpr int A_INSTR = cdivp(A_USED, 64), A_SLOTS = A_INSTR * 64;
  static constexpr int BN = NB * 32;
  static constexpr int B_SLOTS = 18 * BN;                   // [piece 3][tap column 3][group 2][BN]: one kernel row of one chunk
  static constexpr int B_INSTR = B_SLOTS / 64;
  static constexpr int A_BASE0 = 0, B_BASE0 = 2 * A_SLOTS, SPARE = B_BASE0 + 2 * B_SLOTS, TOTAL = SPARE + 64;
  static constexpr int NA = cdivp(A_INSTR, NW), NBI = cdivp(B_INSTR, NW);      // DMA instructions per wave: per chunk (A), per stage (B)
  static constexpr int WM = 8 / WGM, WN = NB / WGN;
  static constexpr int NSTEP = 3 * WM;                      // (tap column, M block) steps of a stage
  static constexpr int NAS = cdivp(NA, 3);                  // A instructions a wave issues per stage (ordinals i with i % 3 == r)
  static constexpr int NDS = NBI + NAS;                     // DMA issue slots per stage
  static constexpr int ISSUE_STEPS = (2 * NSTEP + 2) / 3;   // ... spread over the first two thirds of its steps
  static constexpr size_t LDS_BYTES = (size_t)TOTAL * 16;
  static_assert(B_SLOTS % 64 == 0, "weight stage must be whole DMA instructions");
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");
  static_assert(8 % WGM == 0 && NB % WGN == 0, "wave grid must divide the tile");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <class C>
__global__ __launch_bounds__(C::NT, C::NW == 8 ? 2 : 1) void conv3x3_p3_kernel(P3Args a) {
  constexpr int PLANE = C::PLANE, HWD = C::HWD, BN = C::BN, NW = C::NW, WM = C::WM, WN = C::WN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint4* lds = reinterpret_cast<const uint4*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave / C::WGN, wn = wave % C::WGN;

  int sp, ct;
  if (a.map == 0) {          // Cout tile in the low bits: XCD b % 8 keeps one weight slice hot in its L2
    sp = blockIdx.x / a.nct;
    ct = blockIdx.x % a.nct;
  } else {                   // Cout tiles of one spatial tile on the same XCD: the halo is fetched from HBM once per XCD
    const int j = blockIdx.x >> 3;
    ct = j % a.nct;
    sp = (j / a.nct) * 8 + (blockIdx.x & 7);
    if (sp >= a.nsp) return;
  }
  const int tx = sp % a.tiles_x;
  sp /= a.tiles_x;
  const int ty = sp % a.tiles_y;
  const int n = sp / a.tiles_y;
  const int x0 = tx * C::TW, y0 = ty * C::TH, co0 = ct * BN;
  const int CG = a.Cin >> 3;

  // The DMA instructions go through inline asm: hipcc tracks the builtin as an LDS store and puts s_waitcnt vmcnt(0) in front of the
  // next ds_read, which would serialise the pipeline (conv3x3_bf16_dma.hip).  Ordering is kept by hand: vmcnt(0) + barrier per stage.
  auto make_rsrc = [](const void* p, unsigned bytes) -> i32x4 {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    return i32x4{(int)(unsigned)v, (int)(unsigned)(v >> 32), (int)bytes, 0x00020000};
  };
  const unsigned plane_bytes = (unsigned)a.H * a.W * a.Cin * 2u;
  const i32x4 xrs = make_rsrc(a.x + (size_t)n * 3 * a.H * a.W * a.Cin, 3u * plane_bytes);
  const i32x4 wrs = make_rsrc(a.wpk3, (unsigned)27 * CG * a.CoutP * 16u);
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  auto dma16 = [](const i32x4& rs, unsigned lds_addr, unsigned voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "m0", "memory");
  };
#pragma clang diagnostic pop

  // per-lane source offsets of this wave's DMA instructions (chunk 0, kernel row 0; chunk / row advances ride in the scalar offset)
  unsigned a_off[C::NA], b_off[C::NBI];
#pragma unroll
  for (int i = 0; i < C::NA; ++i) {
    const int e = 64 * (wave + NW * i) + lane;               // slot inside an A buffer
    const int pg = e / PLANE, rem = e % PLANE;
    const int hy = rem / HWD, hx = rem % HWD;
    const int gy = y0 + hy - 1, gx = x0 + hx - 1;
    a_off[i] = (e < C::A_USED && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                   ? (unsigned)(pg >> 1) * plane_bytes + (unsigned)(((gy * a.W + gx) * a.Cin + 8 * (pg & 1)) * 2)
                   : OOB;
  }
#pragma unroll
  for (int i = 0; i < C::NBI; ++i) {
    const int e = 64 * (wave + NW * i) + lane;               // slot inside a B stage buffer
    const int row = e / BN, nn = e % BN;                     // row = (piece * 3 + tap column) * 2 + group
    const int p = row / 6, s = (row % 6) >> 1, g = row & 1;
    b_off[i] = (e < C::B_SLOTS && co0 + nn < a.CoutP) ? (unsigned)((((p * 9 + s) * CG + g) * a.CoutP + co0 + nn) * 16) : OOB;
  }
  const int wv = __builtin_amdgcn_readfirstlane(wave);        // wave-uniform: a scalar for m0
  // i-th A instruction of this wave for chunk kc into A buffer ab / i-th B instruction for (chunk kc, kernel row r) into B buffer bb
  auto dma_a = [&](int i, int kc, int ab, unsigned dead) {
    const int j = wv + NW * i;
    dma16(xrs, lds0 + (unsigned)((j < C::A_INSTR ? ab * C::A_SLOTS + 64 * j : C::SPARE) * 16), a_off[i] | dead, kc * 32);
  };
  auto dma_b = [&](int i, int kc, int r, int bb, unsigned dead) {
    const int j = wv + NW * i;
    dma16(wrs, lds0 + (unsigned)((j < C::B_INSTR ? C::B_BASE0 + bb * C::B_SLOTS + 64 * j : C::SPARE) * 16), b_off[i] | dead,
          (3 * r * CG + 2 * kc) * a.CoutP * 16);
  };

  // fragment addresses.  M block mb = wm * WM + mi: RBW 32: tile row mb; RBW 16: tile rows 2 mb, 2 mb + 1 (second row rotated by 2)
  int a_idx[WM];
  const int pcol = C::RBW == 32 ? li : (li < 16 ? li : ((li - 18) & 15));      // pixel column of this lane inside the tile
  const int prow = C::RBW == 32 ? 0 : (li >> 4);
#pragma unroll
  for (int mi = 0; mi < WM; ++mi) a_idx[mi] = lh * PLANE + ((wm * WM + mi) * C::RBH + prow) * HWD + pcol;
  const int b_idx = lh * BN + wn * WN * 32 + li;

  f32x16 acc[WM][WN];
#pragma unroll
  for (int mi = 0; mi < WM; ++mi)
#pragma unroll
    for (int ni = 0; ni < WN; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int nch_all = a.Cin >> 4;
  const int kc_begin = (int)((long)nch_all * blockIdx.y / a.ksplit);
  const int kc_end = (int)((long)nch_all * (blockIdx.y + 1) / a.ksplit);

  // prologue: the first chunk's A tile and its first kernel row of weights
#pragma unroll
  for (int i = 0; i < C::NA; ++i) dma_a(i, kc_begin, 0, 0u);
#pragma unroll
  for (int i = 0; i < C::NBI; ++i) dma_b(i, kc_begin, 0, 0, 0u);
  __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0)
  __syncthreads();

  int ab = 0, bb = 0;
  for (int kc = kc_begin; kc < kc_end; ++kc) {
    const unsigned dead_a = kc + 1 < kc_end ? 0u : OOB;       // (issued anyway: every wave keeps the same instruction stream; dead lanes land zeros)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      // next stage's weights: (kc, r + 1), or (kc + 1, 0) behind the last kernel row
      const int nkc = r < 2 ? kc : kc + 1, nr = r < 2 ? r + 1 : 0;
      const unsigned dead_b = nkc < kc_end ? 0u : OOB;
      const uint4* As = lds + ab * C::A_SLOTS;
      const uint4* Bs = lds + C::B_BASE0 + bb * C::B_SLOTS;
      uint4 fb[2][3][WN], fa[2][3];
      auto ldB = [&](int s, int set) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int ni = 0; ni < WN; ++ni) fb[set][p][ni] = Bs[b_idx + (p * 3 + s) * 2 * BN + ni * 32];
      };
      auto ldA = [&](int s, int mi, int set) {
#pragma unroll
        for (int p = 0; p < 3; ++p) fa[set][p] = As[a_idx[mi] + p * 2 * PLANE + r * HWD + s];
      };
      ldB(0, 0);
      ldA(0, 0, 0);
#pragma unroll
      for (int s = 0; s < 3; ++s) {
#pragma unroll
        for (int mi = 0; mi < WM; ++mi) {
          const int step = s * WM + mi;
          // DMA issue slots d = 0 .. NDS-1 of this stage, spread over its first steps: first the A instructions of the next chunk
          // (ordinals i = r, r + 3, ...), then the next stage's weights
#pragma unroll
          for (int d = 0; d < C::NDS; ++d) {
            if (d * C::ISSUE_STEPS / C::NDS != step) continue;
            if (d < C::NAS) {
              const int i = r + 3 * d;
              if (i < C::NA) dma_a(i, kc + 1, ab ^ 1, dead_a);
            } else {
              dma_b(d - C::NAS, nkc, nr, bb ^ 1, dead_b);
            }
          }
          if constexpr (C::ILV != 0) __builtin_amdgcn_sched_barrier(0);
          const bool rdA = (mi + 1 < WM) || (s + 1 < 3), rdB = mi == 0 && s + 1 < 3;      // (compile-time after unrolling)
          if (mi + 1 < WM) ldA(s, mi + 1, (step + 1) & 1);
          else if (s + 1 < 3) ldA(s + 1, 0, (step + 1) & 1);
          if (rdB) ldB(s + 1, (s + 1) & 1);
          if constexpr (C::ILV == 0) __builtin_amdgcn_sched_barrier(0);
          const int sa = step & 1, sb = s & 1;
          // pieces: 0 = high, 1 = middle, 2 = low.  Small products first, the dominant hi x hi product last (as conv3x3_f32x3.hip)
          constexpr int PB[6] = {2, 0, 1, 1, 0, 0};
          constexpr int PA[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
          for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int ni = 0; ni < WN; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb[sb][PB[t]][ni]),
                                                                    __builtin_bit_cast(bf16x8_t, fa[sa][PA[t]]), acc[mi][ni], 0, 0, 0);
          if constexpr (C::ILV != 0) {
            if (rdA && rdB) step_groups<6 * WN, 3 + 3 * WN>();
            else if (rdA) step_groups<6 * WN, 3>();
            else if (rdB) step_groups<6 * WN, 3 * WN>();
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);      // this wave's DMA instructions have landed ...
      __syncthreads();                         // ... everybody's have, and everybody is done with this stage's buffers
      bb ^= 1;
    }
    ab ^= 1;
  }

  // ---- epilogue: D = [cout rows][pixel columns]; lane (li, lh) holds its pixel's couts 8 q + 4 lh + (0..3) in registers 4q..4q+3 ----
  const bool split = a.ksplit > 1;       // split-K: raw partial sums, dense [part][n][pixel][Cout]; the epilogue runs in the finalize kernel
  const int cs = split ? a.Cout : a.y_cs;
  const size_t out_elems = (size_t)a.H * a.W * cs;
  void* const anyp = const_cast<uint4*>(a.wpk3);
  float* obase = split ? a.part + ((size_t)blockIdx.y * a.N + n) * out_elems : (a.y != nullptr ? a.y + n * out_elems : nullptr);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(obase != nullptr ? (void*)obase : anyp, 0, obase != nullptr ? (int)(out_elems * 4) : 0, 0x00020000);
  const bool want3 = !split && a.y3 != nullptr;
  const unsigned y3_plane = (unsigned)a.H * a.W * a.y3_cs * 2u;
  const __amdgpu_buffer_rsrc_t y3rs = __builtin_amdgcn_make_buffer_rsrc(want3 ? (void*)(a.y3 + (size_t)n * 3 * a.H * a.W * a.y3_cs) : anyp, 0,
                                                                        want3 ? (int)(3u * y3_plane) : 0, 0x00020000);
  const bool use_mask = !split && a.mask != nullptr;
  const size_t mask_img = (size_t)a.H * a.W * a.mask_cs * (a.mask_p3 ? 3 * 2 : 4);       // bytes per image
  const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(
      use_mask ? (void*)(reinterpret_cast<char*>(const_cast<void*>(a.mask)) + n * mask_img) : anyp, 0,
      use_mask ? (int)((size_t)a.H * a.W * a.mask_cs * (a.mask_p3 ? 2 : 4)) : 0, 0x00020000);      // (P3: plane 0 only)
  const bool use_bias = !split && a.bias != nullptr;
  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(use_bias ? (void*)const_cast<float*>(a.bias) : anyp, 0, use_bias ? a.Cout * 4 : 0, 0x00020000);
  const bool relu = !split && a.relu;
#pragma unroll
  for (int ni = 0; ni < WN; ++ni) {
    const int cblk = co0 + (wn * WN + ni) * 32;
    const int cb = cblk + 4 * lh;
    f32x4 bv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bv[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, (cb + 8 * q) * 4, 0, 0));
#pragma unroll
    for (int mi = 0; mi < WM; ++mi) {
      const int oy = y0 + (wm * WM + mi) * C::RBH + prow, ox = x0 + pcol;
      const bool inside = oy < a.H && ox < a.W;
      const unsigned pixi = (unsigned)(oy * a.W + ox);
      f32x4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = cb + 8 * q;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[q][e] = acc[mi][ni][4 * q + e] + bv[q][e];
          if (relu) v[q][e] = v[q][e] > 0.f ? v[q][e] : 0.f;
        }
        if (use_mask) {
          if (a.mask_p3) {
            typedef short s16x4 __attribute__((ext_vector_type(4)));
            const unsigned moff = (inside && co < a.Cout) ? (pixi * (unsigned)a.mask_cs + (unsigned)co) * 2u : OOB;
            const s16x4 m = __builtin_bit_cast(s16x4, __builtin_amdgcn_raw_buffer_load_b64(mrs, moff, 0, 0));
#pragma unroll
            for (int e = 0; e < 4; ++e) v[q][e] = m[e] > 0 ? v[q][e] : 0.f;
          } else {
            const unsigned moff = (inside && co < a.Cout) ? (pixi * (unsigned)a.mask_cs + (unsigned)co) * 4u : OOB;
            const f32x4 m = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(mrs, moff, 0, 0));
#pragma unroll
            for (int e = 0; e < 4; ++e) v[q][e] = m[e] > 0.f ? v[q][e] : 0.f;
          }
        }
        if (obase != nullptr) {
          const unsigned off = (inside && co < a.Cout) ? (pixi * (unsigned)cs + (unsigned)co) * 4u : OOB;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[q]), yrs, off, 0, 0);
        }
      }
      if (want3) p3_store32(v, y3rs, inside ? pixi * (unsigned)a.y3_cs * 2u : OOB, y3_plane, cblk, lh, a.Cout);
    }
  }
}

// split-K finalize for the P3 convolution: sum of the parts + bias / ReLU / mask, fp32 and / or P3 out.  One thread per (pixel, 8 couts).
__global__ void conv_p3_finalize_kernel(const float* __restrict__ part, const float* __restrict__ bias, const void* __restrict__ mask, int mask_p3,
                                        int mask_cs, float* __restrict__ y, int y_cs, bf16_t* __restrict__ y3, int y3_cs, int N, long hw, int Cout,
                                        int ksplit, int relu) {
  const int c8n = Cout >> 3;
  const long total = (long)N * hw * c8n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c8n) * 8;
    const long np = i / c8n;              // n * hw + pixel
    const long n = np / hw, pix = np % hw;
    f32x4 v[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    for (int k = 0; k < ksplit; ++k) {
      const f32x4* p = reinterpret_cast<const f32x4*>(part + ((size_t)k * N * hw + np) * Cout + c);
      v[0] += p[0];
      v[1] += p[1];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (bias != nullptr) v[h][e] += bias[c + 4 * h + e];
        if (relu) v[h][e] = v[h][e] > 0.f ? v[h][e] : 0.f;
      }
    if (mask != nullptr) {
      if (mask_p3) {
        const unsigned short* m = reinterpret_cast<const unsigned short*>(mask) + ((size_t)n * 3 * hw + pix) * mask_cs + c;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[h][e] = (short)m[4 * h + e] > 0 ? v[h][e] : 0.f;
      } else {
        const float* m = reinterpret_cast<const float*>(mask) + (size_t)np * mask_cs + c;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[h][e] = m[4 * h + e] > 0.f ? v[h][e] : 0.f;
      }
    }
    if (y != nullptr) {
      f32x4* o = reinterpret_cast<f32x4*>(y + (size_t)np * y_cs + c);
      o[0] = v[0];
      o[1] = v[1];
    }
    if (y3 != nullptr) {
      uint2 h0, m0, l0, h1, m1, l1;
      p3_split4(v[0], h0, m0, l0);
      p3_split4(v[1], h1, m1, l1);
      uint4* o = reinterpret_cast<uint4*>(y3 + ((size_t)n * 3 * hw + pix) * y3_cs + c);
      const size_t plane16 = (size_t)hw * y3_cs / 8;       // uint4 per plane
      o[0] = uint4{h0.x, h0.y, h1.x, h1.y};
      o[plane16] = uint4{m0.x, m0.y, m1.x, m1.y};
      o[2 * plane16] = uint4{l0.x, l0.y, l1.x, l1.y};
    }
  }
}

template <class C>
int launch_p3(const P3Args& a0, hipStream_t stream) {
  static bool attr_set_dev[OSVOS_MAX_DEVICES] = {};      // hipFuncSetAttribute is per device
  bool& attr_set = attr_set_dev[osvos_current_device()];
  if (!attr_set) {
    OSVOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_p3_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
    attr_set = true;
  }
  P3Args a = a0;
  a.tiles_x = ceil_div(a.W, C::TW);
  a.tiles_y = ceil_div(a.H, C::TH);
  a.nct = ceil_div(a.CoutP, C::BN);
  a.nsp = a.tiles_x * a.tiles_y * a.N;
  const long blocks = a.map == 0 ? (long)a.nct * a.nsp : (long)a.nct * ((a.nsp + 7) / 8) * 8;
  OSVOS_ARG_CHECK(blocks > 0 && blocks < (1L << 31), "conv3x3 p3: grid of %ld blocks", blocks);
  hipLaunchKernelGGL((conv3x3_p3_kernel<C>), dim3((unsigned)blocks, (unsigned)a.ksplit), dim3(C::NT), C::LDS_BYTES, stream, a);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

struct TileInfoP { int tw, th, bn, nt; size_t lds; };
//                RBW NB WGM WGN
using Q0 = PCfg<32, 4, 4, 2>;   // 32 x 8 px x 128 co, 8 waves (2 x 2 accumulators per wave): two waves per SIMD
using Q1 = PCfg<32, 4, 2, 2>;   // 32 x 8 px x 128 co, 4 waves (4 x 2): one wave per SIMD, 0.375 fragment reads per MFMA
using Q2 = PCfg<32, 2, 4, 2>;   // 32 x 8 px x  64 co, 8 waves (2 x 1)
using Q3 = PCfg<32, 2, 2, 2>;   // 32 x 8 px x  64 co, 4 waves (4 x 1)
using Q4 = PCfg<16, 4, 4, 2>;   // 16 x 16 px x 128 co, 8 waves: narrow maps (107 / 54 pixels wide)
using Q5 = PCfg<16, 2, 4, 2>;   // 16 x 16 px x  64 co, 8 waves
using Q6 = PCfg<16, 2, 2, 2>;   // 16 x 16 px x  64 co, 4 waves
using Q7 = PCfg<32, 1, 8, 1>;   // 32 x 8 px x  32 co, 8 waves (1 x 1): the skinny outputs (side_prep: 16 couts, input gradient: 3)
using Q8 = PCfg<32, 1, 4, 1>;   // 32 x 8 px x  32 co, 4 waves (2 x 1)
using Q9 = PCfg<16, 1, 8, 1>;   // 16 x 16 px x 32 co, 8 waves
using Q10 = PCfg<32, 4, 2, 2, 1>;  // Q1 (4 waves, one per SIMD) with the fragment reads interleaved into the MFMA stream
using Q11 = PCfg<32, 2, 2, 2, 1>;  // Q3 ...
using Q12 = PCfg<16, 2, 2, 2, 1>;  // Q6 ...
using Q13 = PCfg<32, 4, 4, 2, 1>;  // Q0 (8 waves) ...
using Q14 = PCfg<16, 2, 4, 2, 1>;  // Q5 (8 waves) ...
using Q15 = PCfg<32, 2, 4, 2, 1>;  // Q2 (8 waves) ...
constexpr int kNumTilesP = 16;
template <class C>
constexpr TileInfoP infoP() { return TileInfoP{C::TW, C::TH, C::BN, C::NT, C::LDS_BYTES}; }
const TileInfoP kTilesP[kNumTilesP] = {infoP<Q0>(), infoP<Q1>(), infoP<Q2>(), infoP<Q3>(), infoP<Q4>(), infoP<Q5>(), infoP<Q6>(), infoP<Q7>(), infoP<Q8>(), infoP<Q9>(),
                                       infoP<Q10>(), infoP<Q11>(), infoP<Q12>(), infoP<Q13>(), infoP<Q14>(), infoP<Q15>()};

long tiles_of(const TileInfoP& t, int N, int H, int W, int CoutP) {
  return (long)N * ceil_div(H, t.th) * ceil_div(W, t.tw) * ceil_div(CoutP, t.bn);
}

// First rule (to be replaced by measurements, tools/tune_p3.py): the f32x3 rule of conv3x3_f32x3.hip carried over -- 128-cout tiles
// where they still give ~200 workgroups, else 64-cout tiles; 16 x 16 pixel tiles where 32-wide ones pad the frame by more than 10 %.
int pick_tile_p(int N, int H, int W, int CoutP) {
  const bool narrow = (long)ceil_div(W, 32) * 32 * 100 > (long)ceil_div(W, 16) * 16 * 110;
  if (CoutP <= 32) return narrow ? 9 : 7;
  if (CoutP >= 128 && tiles_of(kTilesP[narrow ? 4 : 0], N, H, W, CoutP) >= 200) return narrow ? 4 : 0;
  return narrow ? 5 : 2;
}

int pick_ksplit_p(const TileInfoP& t, int N, int H, int W, int Cin, int CoutP) {
  if (Cin < 256) return 1;
  const long blocks = tiles_of(t, N, H, W, CoutP);
  int ks = 1;
  while (ks < 8 && blocks * ks < 200 && (Cin / 16) / (ks * 2) >= 2) ks *= 2;
  return ks;
}

}  // namespace

int osvos_conv3x3_p3_num_tiles(void) { return kNumTilesP; }

bool osvos_conv3x3_p3_applicable(int Cin, int Cout, int y_cs, int y3_cs) {
  return Cin % 16 == 0 && (y_cs == 0 || (y_cs % 4 == 0 && ((Cout + 3) & ~3) <= y_cs)) && (y3_cs == 0 || (Cout % 8 == 0 && y3_cs % 8 == 0 && y3_cs >= Cout));
}

size_t osvos_conv3x3_p3_splitk_ws_bytes(int N, int H, int W, int Cout) { return (size_t)8 * N * H * W * ((Cout + 3) & ~3) * sizeof(float); }

// x3: P3 activation; wpk3: osvos_pack_x3 pack; y (fp32, channel stride y_cs) and / or y3 (P3, channel stride y3_cs): at least one;
// mask: fp32 NHWC or (mask_p3) a P3 tensor, channel stride mask_cs; tile -1 = automatic (+100: XCD-local halo map); ksplit 0 = automatic
int osvos_conv3x3_p3(const void* x3, const void* wpk3, const float* bias, const void* mask, int mask_p3, int mask_cs, float* y, int y_cs,
                     void* y3, int y3_cs, int N, int H, int W, int Cin, int Cout, int relu, int tile, int ksplit, void* part_ws, hipStream_t stream) {
  OSVOS_ARG_CHECK(x3 && wpk3 && (y || y3), "conv3x3 p3: null pointer");
  OSVOS_ARG_CHECK(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv3x3 p3: bad shape");
  OSVOS_ARG_CHECK(osvos_conv3x3_p3_applicable(Cin, Cout, y ? y_cs : 0, y3 ? y3_cs : 0),
                  "conv3x3 p3: needs Cin %% 16 == 0 (%d); fp32 out: y_cs %% 4 == 0 and >= Cout rounded up to 4; P3 out: Cout %% 8 == 0, y3_cs %% 8 == 0 (%d, %d, %d)",
                  Cin, Cout, y_cs, y3_cs);
  OSVOS_ARG_CHECK(Cout % 4 == 0 || (bias == nullptr && mask == nullptr), "conv3x3 p3: ragged Cout (%d) takes no bias / mask", Cout);
  OSVOS_ARG_CHECK((long)H * W * Cin * 6 < (1L << 31) && (long)H * W * (y ? y_cs : 4) < (1L << 29) && (long)H * W * (y3 ? y3_cs : 8) * 6 < (1L << 31) &&
                      (mask == nullptr || (long)H * W * mask_cs * 4 < (1L << 31)),
                  "conv3x3 p3: image too large for 31-bit byte offsets");
  P3Args a;
  a.x = reinterpret_cast<const bf16_t*>(x3); a.wpk3 = reinterpret_cast<const uint4*>(wpk3); a.bias = bias; a.mask = mask;
  a.mask_p3 = mask_p3 ? 1 : 0; a.mask_cs = mask_cs; a.y = y; a.y3 = reinterpret_cast<bf16_t*>(y3);
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = (Cout + 3) & ~3; a.CoutP = osvos_cout_pad(Cout); a.y_cs = y_cs; a.y3_cs = y3_cs;
  a.relu = relu;
  if (tile < 0) {
    OSVOS_ENV_INT(env_tile, "OSVOS_P3_TILE", -1);
    tile = env_tile >= 0 ? env_tile : pick_tile_p(N, H, W, a.CoutP);
    if (env_tile < 0 && (double)H * W * Cin > 9.0 * Cin * a.CoutP) tile += 100;      // activations larger than the weights
  }
  a.map = tile >= 100 ? 1 : 0;
  tile %= 100;
  OSVOS_ARG_CHECK(tile >= 0 && tile < kNumTilesP, "conv3x3 p3: unknown tile config %d", tile);
  a.part = reinterpret_cast<float*>(part_ws);
  a.ksplit = 1;
  if (part_ws != nullptr) {
    OSVOS_ENV_INT(env_ks, "OSVOS_P3_KSPLIT", 0);
    a.ksplit = ksplit > 0 ? ksplit : (env_ks > 0 && Cin >= 256 ? env_ks : pick_ksplit_p(kTilesP[tile], N, H, W, Cin, a.CoutP));
    if (a.ksplit < 1 || a.ksplit > 8 || a.ksplit > (Cin >> 4) || Cout % 8 != 0) a.ksplit = 1;
  }
  int rc;
  switch (tile) {
    case 0: rc = launch_p3<Q0>(a, stream); break;
    case 1: rc = launch_p3<Q1>(a, stream); break;
    case 2: rc = launch_p3<Q2>(a, stream); break;
    case 3: rc = launch_p3<Q3>(a, stream); break;
    case 4: rc = launch_p3<Q4>(a, stream); break;
    case 5: rc = launch_p3<Q5>(a, stream); break;
    case 6: rc = launch_p3<Q6>(a, stream); break;
    case 7: rc = launch_p3<Q7>(a, stream); break;
    case 8: rc = launch_p3<Q8>(a, stream); break;
    case 9: rc = launch_p3<Q9>(a, stream); break;
    case 10: rc = launch_p3<Q10>(a, stream); break;
    case 11: rc = launch_p3<Q11>(a, stream); break;
    case 12: rc = launch_p3<Q12>(a, stream); break;
    case 13: rc = launch_p3<Q13>(a, stream); break;
    case 14: rc = launch_p3<Q14>(a, stream); break;
    case 15: rc = launch_p3<Q15>(a, stream); break;
    default: osvos_set_error("conv3x3 p3: unknown tile config %d", tile); return -1;
  }
  if (rc || a.ksplit == 1) return rc;
  const long hw = (long)H * W;
  long blocks = ((long)N * hw * (Cout >> 3) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(conv_p3_finalize_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a.part, bias, mask, a.mask_p3, mask_cs, y, y_cs,
                     reinterpret_cast<bf16_t*>(y3), y3_cs, N, hw, Cout, a.ksplit, relu);
  OSVOS_LAUNCH_CHECK();
  return 0;
}
