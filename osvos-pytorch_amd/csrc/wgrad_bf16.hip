// 3x3 convolution weight gradient with bf16 MFMA operands (dtype OSVOS_F32_BF16MFMA): operands rounded to bf16 (RNE), fp32
// accumulation in v_mfma_f32_32x32x16_bf16, fp32 slabs + the shared deterministic reduce (wgrad_f32.hip).
//
// The reduction index k of D[co][ci] += sum_k dY[k][co] * X[k + tap][ci] is the PIXEL axis, and a bf16 MFMA wants 8 consecutive k per
// lane, while the tensors lie [pixel][channel] in HBM.  Kernels in this file, in the order they were written:
//   wgrad_bf16_kernel<XB>     first form.  XB = 0: fp32 tensors in HBM (the bf16 mode without bf16 storage), XB = 1: bf16 tensors.
//                             Tiles are TRANSPOSED in registers on the way into LDS ([channel][row][x], x contiguous: one lane = one
//                             ds_read_b128 = 8 pixels of one channel; channel rows padded by 16 B).  The horizontal tap shift
//                             (0 / 2 / 4 bytes) would misalign a 16-byte read: ONE aligned 20-byte window is read per (row, k-step) and
//                             the three shifted operands are formed in registers (s = 1: four v_alignbit, s = 0 / 2: selection).
//                             64 couts x 64 cins x 9 taps per workgroup, wave (wc, wi) owns a 32 x 32 x 9 block = 9 accumulators.
//   wgrad_bf16pm_kernel<W>    the default for bf16 tensors (OSVOS_WGRAD_FORM=3): pixel-major LDS tiles gathered with ds_read_b64_tr_b16; W = 8 waves on
//                             128-cout tiles where Cout allows.  OSVOS_WGRAD_FORM=0 runs the first form on bf16 tensors as well.
//   The item-load forms (1 / 2: wgrad_bf16v2_kernel) and the LDS-DMA forms (4 / 5: wgrad_bf16dma_kernel, plain / ping-pong) were measured level
//   with or slower than the default in rounds 1-2 (the chip's clock, not the schedule, sets the rate) and are no longer part of the library:
//   tools/native/wgrad_bf16_forms.inc, built into the probe harness only (tools/native/build.sh, -DOSVOS_WGRAD_ALL_FORMS).
// All forms produce bit-identical weight gradients (same patches, splits and k-order).  Probe builds (-DOSVOS_WGRAD_PROF, -DOSVOS_WGRAD_ABL=n:
// tools/native/) add s_memtime phase marks and timing ablations; the shipped library compiles none of that.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

constexpr int PW = 32, PH = 8, PPIX = PW * PH;      // 256-pixel patches: one staging round + two barriers per 144 MFMAs of every wave
constexpr int BCO = 64, BCI = 64;
constexpr int XROWS = PH + 2, XPITCH = 40;                 // halo rows; 40 px (80 B) per row: 5 groups of 8
constexpr int DY_CSTRIDE = PH * PW * 2 + 16;               // bytes per cout row (+16 B pad -> conflict-free b128 columns)
constexpr int X_CSTRIDE = XROWS * XPITCH * 2 + 16;         // bytes per cin row
constexpr int DY_BYTES = BCO * DY_CSTRIDE, X_BYTES = BCI * X_CSTRIDE;
constexpr int DY_ITEMS = (PPIX / 8) * (BCO / 4);           // (pixel group of 8) x (channel quad): 512
constexpr int X_ITEMS = XROWS * (XPITCH / 8) * (BCI / 4);  // 480
constexpr int NDY = DY_ITEMS / 256, NX = (X_ITEMS + 255) / 256;

struct WbArgs {
  const void* x;         // NHWC: fp32 (XB = 0) or bf16 (XB = 1)
  const void* dy;
  float* slab;
  float* bslab;
  int N, H, W, Cin_s, Cout, Cout_s;
  int npx, npy, npatches, per_split, nco_t, nci_t;
  int map;               // 1: XCD-local order (the channel tiles of one split share an XCD = one L2)
  unsigned long long* prof;   // phase counters, read only by builds with -DOSVOS_WGRAD_PROF (tools/native/wgrad_probe.cpp)
  int dbg;               // PROF builds: 1 = skip the X loads, 2 = skip all loads (wrong results; how much do the streamed bytes cost?)
};

__device__ inline unsigned pack2(float lo, float hi) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  bf2 v;
  v[0] = (__bf16)lo;
  v[1] = (__bf16)hi;
  return __builtin_bit_cast(unsigned, v);
}

// XB = 1: x and dy are already bf16 in HBM -- half the bytes, half the staging registers (two workgroups per CU fit)
template <int XB>
__global__ __launch_bounds__(256) void wgrad_bf16_kernel(WbArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* dYs = smem;
  char* Xs = smem + DY_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wc = wave >> 1, wi = wave & 1;

  // map 1: block b runs on XCD b % 8; give every XCD whole splits, so that the nco_t x nci_t workgroups that stream the same pixels
  // (one split) sit behind the same L2 and the patch data crosses the fabric once instead of once per XCD
  int id = blockIdx.x;
  if (a.map == 1) id = (id & 7) * (gridDim.x >> 3) + (id >> 3);
  const int cit = id % a.nci_t;
  id /= a.nci_t;
  const int cot = id % a.nco_t;
  const int split = id / a.nco_t;
  const int co0 = cot * BCO, ci0 = cit * BCI;
  const int p_begin = split * a.per_split, p_end = min(p_begin + a.per_split, a.npatches);

  // staging items (pixel group of 8, channel quad).  tid bits: [0] quad bit 0, [1:2] group bits 0-1, [3:5] quad
  // bits 1-3, [6:7] group bits 2-3.  Eight consecutive lanes = 2 quads x 4 pixel groups: their transposing
  // ds_write_b128 land on 8 distinct 16-byte slots (a plain quad-major order is a 4-way bank conflict -- PMC:
  // 68 % of LDS cycles), while a wave still reads 4 x 256 contiguous bytes per global load instruction.
  const int sq = (tid & 1) | ((tid >> 2) & 14);          // channel quad 0..15 (dY and X tiles are both 64 channels)
  const int sg = ((tid >> 1) & 3) | ((tid >> 4) & 12);   // pixel group 0..15
  const int dq = sq, xq = sq;
  // staging registers: 8 pixels x 4 channels per item -- float4 per pixel (fp32 source) or 4 bf16 = uint2 (bf16 source)
  typedef typename std::conditional<XB != 0, uint2, f32x4>::type stage_t;
  stage_t rdy[NDY][8], rx[NX][8];
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  const bool want_bias = a.bslab != nullptr && cit == 0;
  // Staging through raw buffer loads (as in the convolution kernels): the byte offset of every item relative to the patch origin is
  // computed once; per patch a scalar origin offset is added, columns outside the image are pushed out of range by a compare +
  // select, rows above / below fall out of the per-image buffer range by themselves.  No branch per load (48 of them per patch).
  constexpr unsigned OOB = 0x80000000u;
  constexpr int ES = XB ? 2 : 4;                             // bytes per element
  unsigned dy_rel[NDY], x_rel[NX];
  int dy_c0[NDY], x_c0[NX];                                  // first column of the item inside the patch / halo, -1 = no such item
#pragma unroll
  for (int u = 0; u < NDY; ++u) {
    const int grp = sg + 16 * u;                             // row = grp / 4, x group = grp % 4
    const int co = co0 + 4 * dq;
    dy_rel[u] = (unsigned)((((grp >> 2) * a.W + (grp & 3) * 8) * a.Cout_s + co) * ES);
    dy_c0[u] = co < a.Cout ? (grp & 3) * 8 : -1;
  }
#pragma unroll
  for (int u = 0; u < NX; ++u) {
    const int grp = sg + 16 * u;                             // halo row = grp / 5, x group = grp % 5
    const int hy = grp / 5, hg = grp % 5, ci = ci0 + 4 * xq;
    x_rel[u] = (unsigned)(((hy * a.W + hg * 8) * a.Cin_s + ci) * ES);
    x_c0[u] = (grp < XROWS * 5 && ci < a.Cin_s) ? hg * 8 : -1;
  }
  auto ldb = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned off) -> stage_t {
    if constexpr (XB != 0) {
      typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
      const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0);
      return uint2{v[0], v[1]};
    } else {
      return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    }
  };
  const int img_dy_bytes = a.H * a.W * a.Cout_s * ES, img_x_bytes = a.H * a.W * a.Cin_s * ES;
  auto load_patch = [&](int p) {
    const int px = p % a.npx;
    int t = p / a.npx;
    const int py = t % a.npy;
    const int n = t / a.npy;
    const int x0 = px * PW, y0 = py * PH;
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a.dy)) + (size_t)n * img_dy_bytes, 0, img_dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a.x)) + (size_t)n * img_x_bytes, 0, img_x_bytes, 0x00020000);
    const unsigned dy_base = (unsigned)((y0 * a.W + x0) * a.Cout_s * ES);
    const unsigned x_base = (unsigned)(((y0 - 1) * a.W + (x0 - 1)) * a.Cin_s * ES);      // may be "negative": wraps out of range
#pragma unroll
    for (int u = 0; u < NDY; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned off = (dy_c0[u] >= 0 && x0 + dy_c0[u] + j < a.W) ? dy_rel[u] + dy_base + (unsigned)(j * a.Cout_s * ES) : OOB;
        rdy[u][j] = ldb(drs, off);
      }
#pragma unroll
    for (int u = 0; u < NX; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool ok = x_c0[u] >= 0 && x_c0[u] + j < PW + 2 && (unsigned)(x0 - 1 + x_c0[u] + j) < (unsigned)a.W;
        const unsigned off = ok ? x_rel[u] + x_base + (unsigned)(j * a.Cin_s * ES) : OOB;
        rx[u][j] = ldb(xrs, off);
      }
  };
  // channel c of the pixel pair (j, j+1) -> one dword of two bf16: cvt_pk from fp32, v_perm_b32 byte select from bf16
  auto pair = [&](const stage_t& lo, const stage_t& hi, int c) -> unsigned {
    if constexpr (XB != 0) {
      const unsigned l = c < 2 ? lo.x : lo.y, h = c < 2 ? hi.x : hi.y;
      return __builtin_amdgcn_perm(h, l, (c & 1) ? 0x07060302u : 0x05040100u);
    } else {
      return pack2(lo[c], hi[c]);
    }
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int u = 0; u < NDY; ++u) {
      const int grp = sg + 16 * u;
      if constexpr (XB != 0) {
        if (want_bias) {                                  // bias gradient: fp32 column sums of the (bf16) dY
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            bsum[0] += __uint_as_float(rdy[u][j].x << 16); bsum[1] += __uint_as_float(rdy[u][j].x & 0xffff0000u);
            bsum[2] += __uint_as_float(rdy[u][j].y << 16); bsum[3] += __uint_as_float(rdy[u][j].y & 0xffff0000u);
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum += rdy[u][j];        // bias gradient: exact fp32 column sums of dY
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        uint4 v;
        v.x = pair(rdy[u][0], rdy[u][1], c);
        v.y = pair(rdy[u][2], rdy[u][3], c);
        v.z = pair(rdy[u][4], rdy[u][5], c);
        v.w = pair(rdy[u][6], rdy[u][7], c);
        *reinterpret_cast<uint4*>(dYs + (4 * dq + c) * DY_CSTRIDE + grp * 16) = v;       // [co][row*32 + x]
      }
    }
#pragma unroll
    for (int u = 0; u < NX; ++u) {
      const int grp = sg + 16 * u;
      if (grp < XROWS * 5) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          uint4 v;
          v.x = pair(rx[u][0], rx[u][1], c);
          v.y = pair(rx[u][2], rx[u][3], c);
          v.z = pair(rx[u][4], rx[u][5], c);
          v.w = pair(rx[u][6], rx[u][7], c);
          *reinterpret_cast<uint4*>(Xs + (4 * xq + c) * X_CSTRIDE + grp * 16) = v;        // [ci][hrow*40 + hx]
        }
      }
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const char* a_base = dYs + (wc * 32 + li) * DY_CSTRIDE + lh * 16;
  const char* b_base = Xs + (wi * 32 + li) * X_CSTRIDE + lh * 16;

#ifdef OSVOS_WGRAD_PROF
  unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp = __builtin_amdgcn_s_memtime(), tq;
  const unsigned long long t_begin = tp;
#define WPROF(k) do { __builtin_amdgcn_sched_barrier(0); tq = __builtin_amdgcn_s_memtime(); pt[k] += tq - tp; tp = tq; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define WPROF(k) do { } while (0)
#endif
  if (p_begin < p_end) load_patch(p_begin);
  WPROF(0);
  for (int p = p_begin; p < p_end; ++p) {
    __syncthreads();
    WPROF(1);
#ifdef OSVOS_WGRAD_PROF
    __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0): the wait for the patch's global loads, separated from the transposing stores
    WPROF(2);
#endif
    store_patch();
    WPROF(3);
    __syncthreads();
    WPROF(4);
    if (p + 1 < p_end) load_patch(p + 1);
    WPROF(5);
    // k-step = 16 consecutive pixels of one patch row.  One wave per SIMD (288 registers), so the loop is software
    // pipelined by hand: the 7 LDS reads of k-step ks+1 are requested before the 9 MFMAs of k-step ks issue
    // (two named register sets, sched_barrier pins the order -- hipcc otherwise sinks the reads to their first use)
    struct Frag { uint4 a0; uint4 w0[3]; unsigned w4[3]; };
    auto ldk = [&](int ks, Frag& f) {
      const int row = ks >> 1, kx = ks & 1;
      f.a0 = *reinterpret_cast<const uint4*>(a_base + (row * PW + kx * 16) * 2);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const char* src = b_base + ((row + r) * XPITCH + kx * 16) * 2;
        f.w0[r] = *reinterpret_cast<const uint4*>(src);
        f.w4[r] = reinterpret_cast<const uint4*>(src + 16)->x;
      }
    };
    auto mm = [&](const Frag& f) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        uint4 b[3];
        b[0] = f.w0[r];
        b[1].x = __builtin_amdgcn_alignbit(f.w0[r].y, f.w0[r].x, 16);
        b[1].y = __builtin_amdgcn_alignbit(f.w0[r].z, f.w0[r].y, 16);
        b[1].z = __builtin_amdgcn_alignbit(f.w0[r].w, f.w0[r].z, 16);
        b[1].w = __builtin_amdgcn_alignbit(f.w4[r], f.w0[r].w, 16);
        b[2].x = f.w0[r].y; b[2].y = f.w0[r].z; b[2].z = f.w0[r].w; b[2].w = f.w4[r];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const bf16x8_t bb = __builtin_bit_cast(bf16x8_t, b[s]);
          acc[r * 3 + s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bb, __builtin_bit_cast(bf16x8_t, f.a0), acc[r * 3 + s], 0, 0, 0);
        }
      }
    };
    Frag f0, f1;
    ldk(0, f0);
#pragma unroll 1
    for (int ks = 0; ks < PH * 2; ks += 2) {
      ldk(ks + 1, f1);
      __builtin_amdgcn_sched_barrier(0);
      mm(f0);
      __builtin_amdgcn_sched_barrier(0);
      ldk((ks + 2) & (PH * 2 - 1), f0);                    // unconditional (wraps at the end): the waits stay counted
      __builtin_amdgcn_sched_barrier(0);
      mm(f1);
      __builtin_amdgcn_sched_barrier(0);
    }
    WPROF(6);
  }
  __syncthreads();

  // slab epilogue: the X fragment is the first MFMA operand -> D = [cin rows][cout columns]; lane (li, lh) holds cout li and
  // cins 8 q + 4 lh + (0..3) per register quad = one 16-byte piece of the [tap][co][ci] slab (raw buffer stores, no branches)
  {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const size_t slab_elems = (size_t)9 * a.Cout * a.Cin_s;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(a.slab + (size_t)split * slab_elems, 0, (int)(slab_elems * 4), 0x00020000);
    const int co = co0 + wc * 32 + li;
    const int cib = ci0 + wi * 32 + 4 * lh;
    const unsigned row = co < a.Cout ? (unsigned)(co * a.Cin_s) * 4u : 0x80000000u;
    const unsigned tap_stride = (unsigned)(a.Cout * a.Cin_s) * 4u;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ci = cib + 8 * q;
        const unsigned off = ci < a.Cin_s ? row + (unsigned)t * tap_stride + (unsigned)ci * 4u : 0x80000000u;
        const f32x4 v = {acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), srs, off, 0, 0);
      }
  }
#ifdef OSVOS_WGRAD_PROF
  WPROF(7);
  if (a.prof != nullptr && lane == 0) {
    unsigned long long* q = a.prof + ((size_t)blockIdx.x * 4 + wave) * 10;
    for (int k = 0; k < 8; ++k) q[k] = pt[k];
    q[8] = t_begin;
    q[9] = tp;
  }
#endif
  if (a.bslab != nullptr && cit == 0) {
    f32x4* red = reinterpret_cast<f32x4*>(smem);          // [16 pixel groups][16 quads]
    red[sg * 16 + sq] = bsum;
    __syncthreads();
    if (tid < 16) {
      f32x4 s = red[tid];
#pragma unroll
      for (int m = 1; m < 16; ++m) s += red[m * 16 + tid];
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (co0 + 4 * tid + c < a.Cout) a.bslab[(size_t)split * a.Cout + co0 + 4 * tid + c] = s[c];
    }
  }
}


#ifdef OSVOS_WGRAD_ALL_FORMS      // forms 1 / 2 (probe builds only: tools/native/wgrad_bf16_forms.inc)
#define OSVOS_WGRAD_FORMS_PART 1
#include "../../tools/native/wgrad_bf16_forms.inc"
#undef OSVOS_WGRAD_FORMS_PART
#endif

// ---------------------------------------------------------------------------------------------------------------------------------
// OSVOS_WGRAD_FORM=3 (the default since round 2: bit-identical to the other forms, 1.6 % faster on configs[2]): pixel-major tiles.  The tiles stay the way they lie in HBM -- [pixel][channel], one
// 16-byte piece per lane in, one ds_write_b128 out, no register transposition -- and the MFMA k-fragments (8 consecutive PIXELS of one
// channel per lane) are gathered by the LDS itself: ds_read_b64_tr_b16 hands lane i of a 16-lane group column i of the 4 x 16 element
// block its lanes address (tools/native/tr_probe; lane (i, g) points at pixel i/4, channels 4 (i%4) + 16 g).  A tap shift is then a plain
// address offset: no 20-byte window, no v_alignbit, no register copies -- the k-loop is 20 LDS reads + 9 MFMAs per k-step and nothing
// else.  Pixel pitch = channel bytes + 64 so that the four pixel rows of one read fall on disjoint quarters of the 64 banks.
// Same k-order as the other forms (k = pixel 8 (lane>>5) + element): bit-identical results.
template <int WAVES>
struct PM {
  static constexpr int BCOT = 16 * WAVES, OCT = BCOT / 8, NT = 64 * WAVES;
  static constexpr int DYP = BCOT * 2 + 64, XP = BCI * 2 + 64;            // pixel pitches in bytes: 192 / 320 and 192
  static constexpr int HW_ = PW + 2, XPIX = (PH + 2) * HW_;                 // 34-pixel halo rows, 340 halo pixels
  static constexpr int DY_B = PPIX * DYP, X_B = XPIX * XP;
  static constexpr int NDYL = PH, NXL = (XPIX * 8 + NT - 1) / NT, NLD = NDYL + NXL;      // 16-byte loads per thread and patch: 8 + 11 / 8 + 6
  static constexpr size_t LDS = (size_t)DY_B + X_B;
};

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void wgrad_bf16pm_kernel(WbArgs a) {
  using G = PM<WAVES>;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* dYs = smem;
  char* Xs = smem + G::DY_B;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wc = wave >> 1, wi = wave & 1;
  constexpr unsigned OOB = 0x80000000u;

  int id = blockIdx.x;
  if (a.map == 1) id = (id & 7) * (gridDim.x >> 3) + (id >> 3);
  const int cit = id % a.nci_t;
  id /= a.nci_t;
  const int cot = id % a.nco_t;
  const int split = id / a.nco_t;
  const int co0 = cot * G::BCOT, ci0 = cit * BCI;
  const int p_begin = split * a.per_split, p_end = min(p_begin + a.per_split, a.npatches);

  // staging pieces (16 bytes = 8 channels of one pixel).  dY piece j of this thread: patch row j, column dx, channel octet doct;
  // X piece j: halo pixel xh0 + (NT / 8) j (row-major over the 10 x 34 halo), channel octet xoct
  const int doct = tid % G::OCT, dx = tid / G::OCT;
  const int xoct = tid & 7, xh0 = tid >> 3;
  const bool dy_ch_ok = co0 + 8 * doct < a.Cout, x_ch_ok = ci0 + 8 * xoct < a.Cin_s;
  u32x4 rdy[G::NDYL], rx[G::NXL];
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool want_bias = a.bslab != nullptr && cit == 0;
  const int img_dy_bytes = a.H * a.W * a.Cout_s * 2, img_x_bytes = a.H * a.W * a.Cin_s * 2;
  const unsigned dy_row = (unsigned)(a.W * a.Cout_s * 2);
  const unsigned dy_rel = (unsigned)((dx * a.Cout_s + co0 + 8 * doct) * 2);

  struct Patch { __amdgpu_buffer_rsrc_t drs, xrs; unsigned dy_base, x_base; int x0; };
  auto locate = [&](int p, bool live) -> Patch {
    const int px = p % a.npx;
    int t = p / a.npx;
    const int py = t % a.npy;
    const int n = live ? t / a.npy : 0;
    Patch q;
    q.x0 = live ? px * PW : 0x40000000;            // dead patch: every column test fails -> all loads out of range
    const int y0 = py * PH;
    q.drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(a.dy)) + (size_t)n * img_dy_bytes, 0, img_dy_bytes, 0x00020000);
    q.xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(a.x)) + (size_t)n * img_x_bytes, 0, img_x_bytes, 0x00020000);
    q.dy_base = (unsigned)((y0 * a.W + px * PW) * a.Cout_s * 2);
    q.x_base = (unsigned)(((y0 - 1) * a.W + (px * PW - 1)) * a.Cin_s * 2);       // may be "negative": wraps out of range
    return q;
  };
  auto issue = [&](const Patch& q, int i) {
    if (i >= G::NLD) return;
    if (i < G::NDYL) {
      const unsigned off = (dy_ch_ok && q.x0 + dx < a.W) ? dy_rel + q.dy_base + (unsigned)i * dy_row : OOB;
      rdy[i] = __builtin_amdgcn_raw_buffer_load_b128(q.drs, off, 0, 0);
    } else {
      const int j = i - G::NDYL;
      const int hp = xh0 + (G::NT / 8) * j, hy = hp / G::HW_, hx = hp - hy * G::HW_;
      const bool ok = x_ch_ok && hp < G::XPIX && (unsigned)(q.x0 - 1 + hx) < (unsigned)a.W;
      const unsigned off = ok ? q.x_base + (unsigned)(((hy * a.W + hx) * a.Cin_s + ci0 + 8 * xoct) * 2) : OOB;
      rx[j] = __builtin_amdgcn_raw_buffer_load_b128(q.xrs, off, 0, 0);
    }
  };
  auto store_patch = [&]() {
    if (want_bias) {                                  // bias gradient: fp32 column sums of the (bf16) dY
#pragma unroll
      for (int j = 0; j < G::NDYL; ++j)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          bsum[2 * d] += __uint_as_float(rdy[j][d] << 16);
          bsum[2 * d + 1] += __uint_as_float(rdy[j][d] & 0xffff0000u);
        }
    }
#pragma unroll
    for (int j = 0; j < G::NDYL; ++j)
      *reinterpret_cast<u32x4*>(dYs + (j * PW + dx) * G::DYP + doct * 16) = rdy[j];
#pragma unroll
    for (int j = 0; j < G::NXL; ++j) {
      const int hp = xh0 + (G::NT / 8) * j;
      if (hp < G::XPIX) *reinterpret_cast<u32x4*>(Xs + hp * G::XP + xoct * 16) = rx[j];
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // fragment gather: lane (i = lane & 15, g = (lane >> 4) & 1, lh = lane >> 5) addresses pixel 8 lh + i / 4 (+4 for the second read),
  // channels 16 g + 4 (i % 4) .. +3 of its wave's 32-channel block, and receives channel 16 g + i of pixels 8 lh .. 8 lh + 7
  const int fi = lane & 15, fg = (lane >> 4) & 1, lh = lane >> 5;
  const char* a_base = dYs + (8 * lh + (fi >> 2)) * G::DYP + (32 * wc + 16 * fg + 4 * (fi & 3)) * 2;
  const char* b_base = Xs + (8 * lh + (fi >> 2)) * G::XP + (32 * wi + 16 * fg + 4 * (fi & 3)) * 2;
  auto tr8 = [&](const char* p, int pitch) -> s16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * pitch));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };

#ifdef OSVOS_WGRAD_PROF
  unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp = __builtin_amdgcn_s_memtime(), tq;
  const unsigned long long t_begin = tp;
#endif
  {
    const Patch q = locate(p_begin, p_begin < p_end);
#pragma unroll
    for (int i = 0; i < G::NLD; ++i) issue(q, i);
  }
  WPROF(0);
  for (int p = p_begin; p < p_end; ++p) {
    __syncthreads();
    WPROF(1);
#ifdef OSVOS_WGRAD_PROF
    __builtin_amdgcn_s_waitcnt(0x0f70);
    WPROF(2);
#endif
#if defined(OSVOS_WGRAD_ABL) && OSVOS_WGRAD_ABL == 4
    if (p == p_begin)                                   // ablation 4: the tiles are stored once, not once per patch
#endif
    store_patch();
    WPROF(3);
    __syncthreads();
    WPROF(4);
    const Patch nx = locate(p + 1, p + 1 < p_end);
    WPROF(5);
    // 48 stages = 16 k-steps x 3 tap rows; stage st multiplies tap row r = st % 3 of k-step ks = st / 3 while the three B fragments
    // of stage st + 1 (and, every third stage, the A fragment of the next k-step) are being gathered
    constexpr int NB = WAVES == 8 ? 1 : 2;      // fragment sets: two waves per SIMD cover each other's LDS latency, and 256 registers leave no room for two
    s16x8 af[NB], bfr[NB][3];
    auto lda = [&](int ks) {
#if defined(OSVOS_WGRAD_ABL) && OSVOS_WGRAD_ABL == 2
      if (ks > 1) return;                                  // ablation 2: no LDS fragment reads after the first stages
#endif
      af[ks & (NB - 1)] = tr8(a_base + ((ks >> 1) * PW + (ks & 1) * 16) * G::DYP, G::DYP);
    };
    auto ldb = [&](int st) {
      const int ks = st / 3, r = st % 3;
#if defined(OSVOS_WGRAD_ABL) && OSVOS_WGRAD_ABL == 2
      if (st > 1) return;
#endif
#pragma unroll
      for (int s2 = 0; s2 < 3; ++s2)
        bfr[st & (NB - 1)][s2] = tr8(b_base + (((ks >> 1) + r) * G::HW_ + (ks & 1) * 16 + s2) * G::XP, G::XP);
    };
    if constexpr (NB == 2) {
      lda(0);
      ldb(0);
    }
#pragma unroll
    for (int st = 0; st < PH * 2 * 3; ++st) {
      const int ks = st / 3, r = st % 3;
      if constexpr (NB == 2) {
        if (st + 1 < PH * 2 * 3) {
          if (r == 2) lda(ks + 1);
          ldb(st + 1);
        }
      } else {
        if (r == 0) lda(ks);
        ldb(st);
      }
#if !(defined(OSVOS_WGRAD_ABL) && OSVOS_WGRAD_ABL == 3)                  // ablation 3: no vmem instruction inside the k-loop
      if ((st * G::NLD) / (PH * 6) != ((st + 1) * G::NLD) / (PH * 6)) issue(nx, (st * G::NLD) / (PH * 6));
#endif
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s2 = 0; s2 < 3; ++s2)
#if defined(OSVOS_WGRAD_ABL) && OSVOS_WGRAD_ABL == 1
        acc[r * 3 + s2][0] += __uint_as_float((unsigned)(bfr[st & (NB - 1)][s2][0] ^ bfr[st & (NB - 1)][s2][7] ^ af[ks & (NB - 1)][0] ^ af[ks & (NB - 1)][7]));      // ablation 1: no MFMA
#else
        acc[r * 3 + s2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, bfr[st & (NB - 1)][s2]),
                                                                __builtin_bit_cast(bf16x8_t, af[ks & (NB - 1)]), acc[r * 3 + s2], 0, 0, 0);
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
    WPROF(6);
  }
  __syncthreads();

  {   // slab epilogue, as in the other forms
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
#ifdef OSVOS_WGRAD_PROF
  WPROF(7);
  if (a.prof != nullptr && lane == 0) {
    unsigned long long* q = a.prof + ((size_t)blockIdx.x * WAVES + wave) * 10;
    for (int k = 0; k < 8; ++k) q[k] = pt[k];
    q[8] = t_begin;
    q[9] = tp;
  }
#endif
  if (want_bias) {
    float* red = reinterpret_cast<float*>(smem);          // [32 columns][OCT octets][8 channels]
#pragma unroll
    for (int c = 0; c < 8; ++c) red[(dx * G::OCT + doct) * 8 + c] = bsum[c];
    __syncthreads();
    if (tid < G::BCOT) {
      float sum = 0.f;
      for (int g = 0; g < 32; ++g) sum += red[g * G::BCOT + tid];
      if (co0 + tid < a.Cout) a.bslab[(size_t)split * a.Cout + co0 + tid] = sum;
    }
  }
}

#ifdef OSVOS_WGRAD_ALL_FORMS      // forms 4 / 5 (probe builds only)
#define OSVOS_WGRAD_FORMS_PART 2
#include "../../tools/native/wgrad_bf16_forms.inc"
#undef OSVOS_WGRAD_FORMS_PART
#endif

#ifdef OSVOS_WGRAD_ALL_FORMS
constexpr size_t kLdsV2 = (size_t)DY_BYTES + X_BYTES, kLdsV2w = (size_t)2 * DY_BYTES + X_BYTES;
#endif

constexpr int kDefaultMap = 1;      // bf16-input form: XCD-local split order (L2 hit rate 38 % -> 77 %, HBM reads / 3; OSVOS_WGRAD_MAP=0 turns it off)
constexpr int kDefaultForm = 3;     // bf16-input kernel: 0 = first staging form, 1 = second (wgrad_bf16v2_kernel<4>), 2 = second with eight waves / 128-cout
constexpr int kWideForm = 2;        // tiles where Cout allows (OSVOS_WGRAD_FORM)
unsigned long long* g_wgrad_prof = nullptr;

struct WbPlan {
  int nco_t, nci_t, npx, npy, npatches, nsplit, per_split;
  size_t slab_floats, bslab_floats;
};

// bco = 128: the eight-wave form (one workgroup per CU: aim at one round of 256 workgroups)
WbPlan make_plan(int N, int H, int W, int Cin_s, int Cout, int bco = BCO) {
  WbPlan p;
  p.nco_t = ceil_div(Cout, bco);
  p.nci_t = ceil_div(Cin_s, BCI);
  p.npx = ceil_div(W, PW);
  p.npy = ceil_div(H, PH);
  p.npatches = N * p.npx * p.npy;
  // workgroups aimed at: 512 for the four-wave form (two per CU), 192 for the eight-wave form (one per CU -- round 5: three quarters of the CUs
  // instead of all of them, the data-gradient chain that runs beside it keeps a share: +0.6-1.0 % on configs[2] on two boxes; 128 / 224 / 256 level
  // to -0.8 %, 384 -1.4 %, 512 -3 %; profiles/r05_ab_small.txt)
  int want = ceil_div(bco == BCO ? 512 : 192, p.nco_t * p.nci_t);
  const int max_split = p.npatches / 2 > 0 ? p.npatches / 2 : 1;
  if (want > max_split) want = max_split;
  if (want > 256) want = 256;
  p.per_split = ceil_div(p.npatches, want);
  p.nsplit = ceil_div(p.npatches, p.per_split);
  p.slab_floats = (size_t)p.nsplit * 9 * Cout * Cin_s;
  p.bslab_floats = (size_t)p.nsplit * Cout;
  return p;
}

}  // namespace

int osvos_wgrad_reduce_launch(const float* slab, const float* bslab, float* dw, float* db, int nsplit, int Cout, int Cin,
                              int Cin_s, int accumulate, hipStream_t stream);

// shapes the bf16 kernel takes: the wide trunk layers (Cin_s, Cout multiples of 64); everything else stays on the fp32 kernels
// (Cout = 16, the side_prep layers: one 64-cout tile with 16 live rows -- 75 % of the MFMA rows multiply zeros, still 2x faster than
//  the exact-fp32 skinny kernel, whose cost is staging the wide X tile either way)
bool osvos_wgrad_bf16_applicable(int Cin_s, int Cout) { return Cin_s % 64 == 0 && (Cout % 64 == 0 || Cout == 16); }

size_t osvos_wgrad_bf16_ws_bytes(int N, int H, int W, int Cin_s, int Cout) {
  if (!osvos_wgrad_bf16_applicable(Cin_s, Cout)) return 0;
  const WbPlan p = make_plan(N, H, W, Cin_s, Cout), pw = make_plan(N, H, W, Cin_s, Cout, 128);      // whichever form runs
  const size_t f = p.slab_floats + p.bslab_floats, fw = pw.slab_floats + pw.bslab_floats;
  return align_up((f > fw ? f : fw) * sizeof(float), 256);
}

// xb = 0: x and dy fp32; xb = 1: both bf16
int osvos_conv3x3_wgrad_bf16mfma_io(const void* x, const void* dy, int xb, void* ws, float* dw, float* db,
                                    int N, int H, int W, int Cin, int Cin_s, int Cout, int Cout_s,
                                    int accumulate, hipStream_t stream) {
  OSVOS_ARG_CHECK(x && dy && ws && dw, "wgrad bf16: null pointer");
  OSVOS_ARG_CHECK(osvos_wgrad_bf16_applicable(Cin_s, Cout) && Cin == Cin_s && Cout_s % 4 == 0, "wgrad bf16: unsupported shape");
  OSVOS_ARG_CHECK((long)H * W * Cin_s < (1L << 29) && (long)H * W * Cout_s < (1L << 29), "wgrad bf16: image too large for 31-bit byte offsets");
  static const int form_env = getenv("OSVOS_WGRAD_FORM") ? atoi(getenv("OSVOS_WGRAD_FORM")) : -1;
#ifdef OSVOS_WGRAD_ALL_FORMS
  const int form = xb ? (form_env >= 0 ? form_env : kDefaultForm) : 0;
#else      // the shipped library holds two forms: 0 (first staging form; the only one for fp32 tensors) and 3 (pixel-major, the default for bf16 tensors)
  const int form = xb ? ((form_env == 0) ? 0 : kDefaultForm) : 0;
#endif
  const bool wide = form >= kWideForm && Cout % 128 == 0;        // eight-wave form: 128-cout tiles
  WbPlan p = make_plan(N, H, W, Cin_s, Cout, wide ? 128 : BCO);
  WbArgs a;
  a.x = x; a.dy = dy;
  a.slab = reinterpret_cast<float*>(ws);
  a.bslab = db ? a.slab + p.slab_floats : nullptr;
  a.N = N; a.H = H; a.W = W; a.Cin_s = Cin_s; a.Cout = Cout; a.Cout_s = Cout_s;
  a.npx = p.npx; a.npy = p.npy; a.npatches = p.npatches; a.per_split = p.per_split; a.nco_t = p.nco_t; a.nci_t = p.nci_t;
  const long blocks = (long)p.nsplit * p.nco_t * p.nci_t;
  static const int map_env = getenv("OSVOS_WGRAD_MAP") ? atoi(getenv("OSVOS_WGRAD_MAP")) : -1;
  a.map = (map_env >= 0 ? map_env : kDefaultMap) == 1 && blocks % 8 == 0 ? 1 : 0;
  a.prof = g_wgrad_prof;
  static const int dbg_env = getenv("OSVOS_WGRAD_DBG") ? atoi(getenv("OSVOS_WGRAD_DBG")) : 0;
  a.dbg = dbg_env;
  constexpr size_t lds = (size_t)DY_BYTES + X_BYTES;
  static bool attr_set_dev[OSVOS_MAX_DEVICES] = {};      // hipFuncSetAttribute is per device
  bool& attr_set = attr_set_dev[osvos_current_device()];
  if (!attr_set) {
    OSVOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_bf16_kernel<0>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    OSVOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_bf16_kernel<1>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const int phase = osvos_wgrad_phase();
#ifdef OSVOS_WGRAD_ALL_FORMS
  if (phase != 2 && form >= 4 && wide) {                // LDS-DMA forms: 4 fragments one stage ahead, 5 ping-pong segments
    static bool attr4_set_dev[OSVOS_MAX_DEVICES] = {};
    bool& attr4_set = attr4_set_dev[osvos_current_device()];
    constexpr int lds_dma = (int)DM::LDS;
    if (!attr4_set) {
      OSVOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_bf16dma_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_dma));
      OSVOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_bf16dma_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_dma));
      attr4_set = true;
    }
    if (form == 5) hipLaunchKernelGGL(wgrad_bf16dma_kernel<1>, dim3((unsigned)blocks), dim3(512), lds_dma, stream, a);
    else hipLaunchKernelGGL(wgrad_bf16dma_kernel<0>, dim3((unsigned)blocks), dim3(512), lds_dma, stream, a);
    OSVOS_LAUNCH_CHECK();
  } else
#endif
  if (phase != 2 && form >= 3) {
    static bool attr3_set_dev[OSVOS_MAX_DEVICES] = {};      // per device, like every other kernel attribute
    bool& attr3_set = attr3_set_dev[osvos_current_device()];
    constexpr size_t lds4 = PM<4>::LDS, lds8 = PM<8>::LDS;
    if (!attr3_set) {
      OSVOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_bf16pm_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
      OSVOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_bf16pm_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8));
      attr3_set = true;
    }
    if (wide) hipLaunchKernelGGL(wgrad_bf16pm_kernel<8>, dim3((unsigned)blocks), dim3(512), lds8, stream, a);
    else hipLaunchKernelGGL(wgrad_bf16pm_kernel<4>, dim3((unsigned)blocks), dim3(256), lds4, stream, a);
    OSVOS_LAUNCH_CHECK();
#ifdef OSVOS_WGRAD_ALL_FORMS
  } else if (phase != 2 && form != 0) {
    static bool attr2_set_dev[OSVOS_MAX_DEVICES] = {};
    bool& attr2_set = attr2_set_dev[osvos_current_device()];
    if (!attr2_set) {
      OSVOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_bf16v2_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsV2));
      OSVOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_bf16v2_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsV2w));
      attr2_set = true;
    }
    if (wide) hipLaunchKernelGGL(wgrad_bf16v2_kernel<8>, dim3((unsigned)blocks), dim3(512), kLdsV2w, stream, a);
    else hipLaunchKernelGGL(wgrad_bf16v2_kernel<4>, dim3((unsigned)blocks), dim3(256), kLdsV2, stream, a);
    OSVOS_LAUNCH_CHECK();
#endif
  } else if (phase != 2) {
    if (xb) hipLaunchKernelGGL(wgrad_bf16_kernel<1>, dim3((unsigned)blocks), dim3(256), lds, stream, a);
    else hipLaunchKernelGGL(wgrad_bf16_kernel<0>, dim3((unsigned)blocks), dim3(256), lds, stream, a);
    OSVOS_LAUNCH_CHECK();
  }
  if (phase == 1) return 0;
  return osvos_wgrad_reduce_launch(a.slab, a.bslab, dw, db, p.nsplit, Cout, Cin, Cin_s, accumulate, stream);
}

#ifdef OSVOS_WGRAD_PROF
extern "C" void osvos_debug_set_wgrad_prof_bf16(unsigned long long* p) { g_wgrad_prof = p; }
#endif

int osvos_conv3x3_wgrad_bf16mfma(const float* x, const float* dy, void* ws, float* dw, float* db,
                                 int N, int H, int W, int Cin, int Cin_s, int Cout, int Cout_s,
                                 int accumulate, hipStream_t stream) {
  return osvos_conv3x3_wgrad_bf16mfma_io(x, dy, 0, ws, dw, db, N, H, W, Cin, Cin_s, Cout, Cout_s, accumulate, stream);
}
