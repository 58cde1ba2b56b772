// 3x3 convolution weight + bias gradient, NHWC fp32, on v_mfma_f32_32x32x2_f32 (gfx950).
//
// Replaces the weight/bias half of aten::convolution_backward for nn.Conv2d(k=3, padding=1)
// (reference vgg_osvos.py:41,142; invoked by loss.backward(), train_online.py:141).
//
// GEMM view: D[co][ci] (per tap) = sum over pixels dY[p][co] * X[p + tap][ci]; the reduction
// dimension is the PIXEL axis (up to 409,920 at 854x480) and the output is tiny, so
//   * a workgroup owns a (CB*32 couts) x (IB*32 cins) x 9 taps output tile (one 32x32 MFMA
//     block pair per wave, 9 tap accumulators = 144 accumulator registers) and walks a range of
//     32x2-pixel patches ("split"); per patch the dY tile and the X halo tile are staged once in
//     LDS in their natural [pixel][channel] order -- the MFMA A operand (dY^T) and B operand
//     (X shifted by the tap) are then plain conflict-free ds_read_b32 rows, and the 9 taps
//     re-use the same X tile through an LDS address offset
//   * LDS holds ONE patch; the next patch waits in registers (global loads in flight during the
//     MFMAs), so two workgroups fit per CU and cover each other's barrier/LDS latencies
//   * splits write fp32 partial slabs already in the reference's OIHW order ([co][ci][tap]); a
//     second deterministic, fully coalesced pass sums them and (optionally) accumulates
//   * the bias gradient (column sums of dY) rides along in the workgroups of the first Cin tile
#include "common.h"

namespace {

// pixel patch of 64 output pixels: 32 x 2 by default, 16 x 4 (PWT = 16) where that pads the frame less (107-pixel wide conv4_x:
// 112 instead of 128 columns are multiplied); halo = patch + 1 pixel on every side
constexpr int PPIX = 64;
template <int PWT> struct Geo {
  static constexpr int PW = PWT, PH = PPIX / PWT, XW = PWT + 2, XH = PH + 2, XPIX = XW * XH;
};

struct WgArgs {
  const float* x;
  const float* dy;
  float* slab;
  float* bslab;
  int N, H, W, Cin_s, Cout, Cout_s;
  int npx, npy, npatches, nsplit, per_split;
  int nco_t, nci_t;
  int oihw;      // slab element order: 1 = [co][ci][tap] (reference layout), 0 = [tap][co][ci] (coalesced stores)
};

// PIPE: pinned software pipeline of the operand fetch; DBUF: two LDS patch buffers (one barrier per
// patch) instead of one (two barriers, half the LDS); OCC: __launch_bounds__ waves/SIMD
template <int CB, int IB, int PIPE, int DBUF, int OCC, int PWT = 32>
__global__ __launch_bounds__(256, OCC) void wgrad_f32_kernel(WgArgs a) {
  constexpr int PW = Geo<PWT>::PW, PH = Geo<PWT>::PH, XW = Geo<PWT>::XW, XPIX = Geo<PWT>::XPIX;
  constexpr int BCO = CB * 32, BCI = IB * 32;
  constexpr int DY_F4 = PPIX * BCO / 4, X_F4 = XPIX * BCI / 4;
  constexpr int BUF_F4 = DY_F4 + X_F4;
  constexpr int NDY = (DY_F4 + 255) / 256, NX = (X_F4 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* lds4 = reinterpret_cast<f32x4*>(smem);
  const float* lds = reinterpret_cast<const float*>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int cb = wave / IB, ib = wave % IB;

  int id = blockIdx.x;
  const int cit = id % a.nci_t;
  id /= a.nci_t;
  const int cot = id % a.nco_t;
  const int split = id / a.nco_t;
  const int co0 = cot * BCO, ci0 = cit * BCI;
  const int p_begin = split * a.per_split;
  const int p_end = min(p_begin + a.per_split, a.npatches);

  // Staging through raw buffer loads (as in conv3x3_f32.hip): per lane the byte offset of every item RELATIVE to the patch origin
  // is computed once; per patch only the scalar origin offset, an x-range compare and a select are left.  Rows above / below the
  // image fall out of the per-image buffer range by themselves (offset < 0 wraps, offset >= H*W*C*4), columns need the compare.
  constexpr unsigned OOB = 0x80000000u;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  unsigned dy_rel[NDY], x_rel[NX];
  int dy_dx[NDY], x_dx[NX];                                   // column of the item inside the patch / halo (-1: no such item)
#pragma unroll
  for (int i = 0; i < NDY; ++i) {
    const int e = tid + i * 256;
    const int pix = e / (BCO / 4), q = e % (BCO / 4);
    const bool ok = e < DY_F4 && co0 + 4 * q < a.Cout;
    dy_rel[i] = (unsigned)((((pix / PW) * a.W + pix % PW) * a.Cout_s + co0 + 4 * q) * 4);
    dy_dx[i] = ok ? pix % PW : -1;
  }
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int e = tid + i * 256;
    const int pix = e / (BCI / 4), q = e % (BCI / 4);
    const bool ok = e < X_F4 && ci0 + 4 * q < a.Cin_s;
    x_rel[i] = (unsigned)((((pix / XW) * a.W + pix % XW) * a.Cin_s + ci0 + 4 * q) * 4);
    x_dx[i] = ok ? pix % XW : -1;
  }
  const int img_dy_bytes = a.H * a.W * a.Cout_s * 4, img_x_bytes = a.H * a.W * a.Cin_s * 4;
  u32x4 rdy[NDY], rx[NX];
  auto load_patch = [&](int p) {
    const int px = p % a.npx;
    int t = p / a.npx;
    const int py = t % a.npy;
    const int n = t / a.npy;
    const int x0 = px * PW, y0 = py * PH;
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy) + (size_t)n * a.H * a.W * a.Cout_s, 0, img_dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) + (size_t)n * a.H * a.W * a.Cin_s, 0, img_x_bytes, 0x00020000);
    const unsigned dy_base = (unsigned)((y0 * a.W + x0) * a.Cout_s * 4);
    const unsigned x_base = (unsigned)(((y0 - 1) * a.W + (x0 - 1)) * a.Cin_s * 4);     // may be "negative": wraps out of range
#pragma unroll
    for (int i = 0; i < NDY; ++i) {
      const unsigned off = (dy_dx[i] >= 0 && x0 + dy_dx[i] < a.W) ? dy_rel[i] + dy_base : OOB;
      rdy[i] = __builtin_amdgcn_raw_buffer_load_b128(drs, off, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const unsigned off = (x_dx[i] >= 0 && (unsigned)(x0 - 1 + x_dx[i]) < (unsigned)a.W) ? x_rel[i] + x_base : OOB;
      rx[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0);
    }
  };
  auto store_patch = [&](int buf) {
    f32x4* d = lds4 + buf * BUF_F4;
#pragma unroll
    for (int i = 0; i < NDY; ++i) {
      const int e = tid + i * 256;
      if (e < DY_F4) d[e] = __builtin_bit_cast(f32x4, rdy[i]);
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int e = tid + i * 256;
      if (e < X_F4) d[DY_F4 + e] = __builtin_bit_cast(f32x4, rx[i]);
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;
  const bool do_bias = (a.bslab != nullptr) && (cit == 0);
  constexpr int BG = 256 / BCO;            // pixel groups for the bias column sums
  const int bco = tid % BCO, bgrp = tid / BCO;

  auto compute = [&](int buf) {
    const float* dYs = lds + (size_t)buf * BUF_F4 * 4;
    const float* Xs = dYs + DY_F4 * 4;
    if (PIPE) {
      // software-pipelined over the 32 pixel pairs: the 10 LDS operands of pair pp+1 are requested
      // before the 9 MFMAs of pair pp issue (two named register sets, loop kept rolled: a full
      // unroll spills); sched_barrier pins that order, hipcc otherwise sinks the ds_reads
      float av0, av1, bv0[9], bv1[9];
      auto ld = [&](int pp, float& av, float (&bv)[9]) {
        const int dy = pp / (PW / 2), dx = (pp % (PW / 2)) * 2 + lh;
        av = dYs[(dy * PW + dx) * BCO + cb * 32 + li];
        const float* xb = Xs + (dy * XW + dx) * BCI + ib * 32 + li;
#pragma unroll
        for (int t = 0; t < 9; ++t) bv[t] = xb[((t / 3) * XW + (t % 3)) * BCI];
      };
      ld(0, av0, bv0);
#pragma unroll 1
      for (int pp = 0; pp < PPIX / 2; pp += 2) {
        ld(pp + 1, av1, bv1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv0[t], av0, acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        ld((pp + 2) & (PPIX / 2 - 1), av0, bv0);     // unconditional (wraps at the end): waits stay counted
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv1[t], av1, acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll 2
      for (int pp = 0; pp < PPIX / 2; ++pp) {
        const int dy = pp / (PW / 2), dx = (pp % (PW / 2)) * 2 + lh;
        const float av = dYs[(dy * PW + dx) * BCO + cb * 32 + li];
        const float* xb = Xs + (dy * XW + dx) * BCI + ib * 32 + li;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const float bv = xb[((t / 3) * XW + (t % 3)) * BCI];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, acc[t], 0, 0, 0);
        }
      }
    }
    if (do_bias) {
#pragma unroll 4
      for (int pix = bgrp; pix < PPIX; pix += BG) bsum += dYs[pix * BCO + bco];
    }
  };

  if (DBUF) {
    if (p_begin < p_end) {
      load_patch(p_begin);
      store_patch(0);
    }
    __syncthreads();
    int buf = 0;
    for (int p = p_begin; p < p_end; ++p) {
      const bool more = p + 1 < p_end;
      if (more) load_patch(p + 1);
      compute(buf);
      if (more) store_patch(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  } else {
    if (p_begin < p_end) load_patch(p_begin);
    for (int p = p_begin; p < p_end; ++p) {
      __syncthreads();                 // every wave is done reading the previous patch
      store_patch(0);
      __syncthreads();
      if (p + 1 < p_end) load_patch(p + 1);      // in flight while this patch is multiplied
      compute(0);
    }
    __syncthreads();
  }

  // ---- write the partial slab.  The x fragment is the FIRST MFMA operand, so D = [cin rows][cout columns]: lane
  // (li, lh) holds cout li and, per tap, cins 8 q + 4 lh + (0..3) in registers 4q..4q+3 = one 16-byte piece of the
  // coalesced [tap][co][ci] slab.  Raw buffer stores, offset out of range past Cout / Cin_s: no branches, no 64-bit
  // address arithmetic (144 guarded dword stores per wave before).
  const int co = co0 + cb * 32 + li;
  if (!a.oihw) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const size_t slab_elems = (size_t)9 * a.Cout * a.Cin_s;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(a.slab + (size_t)split * slab_elems, 0, (int)(slab_elems * 4), 0x00020000);
    const int cib = ci0 + ib * 32 + 4 * lh;
    const unsigned row = co < a.Cout ? (unsigned)(co * a.Cin_s) * 4u : 0x80000000u;
    const unsigned tap_stride = (unsigned)(a.Cout * a.Cin_s) * 4u;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ci = cib + 8 * q;
        const unsigned off = ci < a.Cin_s ? row + (unsigned)t * tap_stride + (unsigned)ci * 4u : 0x80000000u;
        f32x4 v = {acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), srs, off, 0, 0);
      }
  } else {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ci0 + ib * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (co < a.Cout && ci < a.Cin_s) a.slab[(((size_t)split * a.Cout + co) * a.Cin_s + ci) * 9 + t] = acc[t][r];
      }
  }
  if (do_bias) {
    float* red = reinterpret_cast<float*>(smem);   // all LDS reads are behind the last barrier
    red[tid] = bsum;
    __syncthreads();
    if (tid < BCO) {
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < BG; ++g) s += red[g * BCO + tid];
      if (co0 + tid < a.Cout) a.bslab[(size_t)split * a.Cout + co0 + tid] = s;
    }
  }
}

// dw[co][ci][tap] (OIHW, Cin real) (+)= sum_split slab[split][...]; one workgroup = 32 float4 columns
// (128 consecutive slab elements) x 8 split lanes: 16-byte coalesced reads with several loads in
// flight per thread, LDS combine, then the transposing write
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bslab,
                                                           float* __restrict__ dw, float* __restrict__ db,
                                                           int nsplit, int Cout, int Cin, int Cin_s, int accumulate, int oihw) {
  __shared__ f32x4 red[256];
  const int total = Cout * Cin_s * 9;                 // multiple of 4 (Cin_s % 4 == 0)
  const int e = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int idx = (blockIdx.x * 32 + e) * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (idx < total) {
    const f32x4* src = reinterpret_cast<const f32x4*>(slab + idx);
    const size_t stride4 = (size_t)total / 4;
    int sp = sl;
    for (; sp + 24 < nsplit; sp += 32) {              // 4 independent loads in flight
      const f32x4 a = src[(size_t)sp * stride4], b = src[(size_t)(sp + 8) * stride4];
      const f32x4 c = src[(size_t)(sp + 16) * stride4], d = src[(size_t)(sp + 24) * stride4];
      s += (a + b) + (c + d);
    }
    for (; sp < nsplit; sp += 8) s += src[(size_t)sp * stride4];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (sl == 0 && idx < total) {
#pragma unroll
    for (int k = 1; k < 8; ++k) s += red[k * 32 + e];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = idx + q;
      int t, ci, co;
      if (oihw) { t = i % 9; ci = (i / 9) % Cin_s; co = i / (9 * Cin_s); }
      else { ci = i % Cin_s; co = (i / Cin_s) % Cout; t = i / (Cin_s * Cout); }
      if (ci < Cin) {
        float* o = dw + ((size_t)co * Cin + ci) * 9 + t;
        *o = accumulate ? (*o + s[q]) : s[q];
      }
    }
  }
  // bias gradient: the workgroups past the weight columns take 32 couts x 8 split lanes each (the first version let ONE
  // workgroup walk all splits serially: 256 dependent-latency loads that set the kernel's duration on the small layers)
  const int nblk_dw = (total / 4 + 31) / 32;
  if (db != nullptr && (int)blockIdx.x >= nblk_dw) {
    const int co = ((int)blockIdx.x - nblk_dw) * 32 + e;
    float b = 0.f;
    if (co < Cout) {
      int sp = sl;
      for (; sp + 24 < nsplit; sp += 32)
        b += (bslab[(size_t)sp * Cout + co] + bslab[(size_t)(sp + 8) * Cout + co]) + (bslab[(size_t)(sp + 16) * Cout + co] + bslab[(size_t)(sp + 24) * Cout + co]);
      for (; sp < nsplit; sp += 8) b += bslab[(size_t)sp * Cout + co];
    }
    __syncthreads();                       // (red[] above is only written by weight-column workgroups; this one reuses it)
    red[threadIdx.x][0] = b;
    __syncthreads();
    if (sl == 0 && co < Cout) {
#pragma unroll
      for (int k = 1; k < 8; ++k) b += red[k * 32 + e][0];
      db[co] = accumulate ? (db[co] + b) : b;
    }
  }
}

// Slab reduce of the wide layers (Cin_s a multiple of 64), transposing through LDS: a workgroup owns ONE cout x 64 cins x 9 taps.
// Wave w sums the splits w, w + 4, ... for all 9 taps with lanes = cins (256-byte coalesced slab reads), the four partial sets meet in
// LDS, and the 576 results leave as ONE contiguous run of dw (OIHW: [co][ci][tap], 2304 bytes) -- the kernel above writes the same
// values as 4-byte stores 36 bytes apart (47 us for 37.7 MB of slabs at batch 1, 90 us beside full-chip kernels at batch 12: 1.45 ms of
// the 13 ms bf16 parent step).  Workgroups of the first cin block also reduce their cout's bias partials.
// NW waves per workgroup: 4 where the grid is large (deep layers: few splits, thousands of workgroups), 16 where it is not -- conv1_2 has
// 256 splits and ONE channel tile pair: 64 workgroups each walking 64 x 9 dependent 256-byte loads per wave took 46 us at the very end
// of the step (the last slab reduce is exposed: nothing is left to run beside it).
// (A float4-load form with 18 KB in flight per wave and an in-place fold pass for the many-split layers was built and dropped in round 3:
//  beside the matrix kernels it always shares the chip with, its launches took as long as these and the step was 2 % slower -- what a
//  side-stream bandwidth kernel costs is the CU time it holds, not its stand-alone bandwidth; profiles/r03_ab_glue_kernels.txt.)
template <int NW>
__global__ __launch_bounds__(64 * NW) void wgrad_reduce_t_kernel(const float* __restrict__ slab, const float* __restrict__ bslab,
                                                                 float* __restrict__ dw, float* __restrict__ db,
                                                                 int nsplit, int Cout, int Cin_s, int accumulate) {
  __shared__ float red[NW][9][64];
  __shared__ float outb[576];
  constexpr int NT = 64 * NW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ci0 = blockIdx.x * 64, co = blockIdx.y;
  const size_t tap_stride = (size_t)Cout * Cin_s, split_stride = 9 * tap_stride;
  const float* src = slab + (size_t)co * Cin_s + ci0 + lane;
  float acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = 0.f;
  for (int sp = wave; sp < nsplit; sp += NW) {
    const float* q = src + (size_t)sp * split_stride;
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] += q[(size_t)t * tap_stride];
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) red[wave][t][lane] = acc[t];
  __syncthreads();
  for (int e = threadIdx.x; e < 576; e += NT) {
    const int t = e / 64, ci = e % 64;
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < NW; g += 4) v += (red[g][t][ci] + red[g + 1][t][ci]) + (red[g + 2][t][ci] + red[g + 3][t][ci]);      // fixed order
    outb[ci * 9 + t] = v;
  }
  __syncthreads();
  float* dst = dw + ((size_t)co * Cin_s + ci0) * 9;        // Cin == Cin_s for these layers
  for (int e = threadIdx.x; e < 576; e += NT) dst[e] = accumulate ? dst[e] + outb[e] : outb[e];
  if (db != nullptr && blockIdx.x == 0) {
    float b = 0.f;
    for (int sp = threadIdx.x; sp < nsplit; sp += NT) b += bslab[(size_t)sp * Cout + co];
    b = wave_sum(b);
    __syncthreads();
    if (lane == 0) red[0][0][wave] = b;
    __syncthreads();
    if (threadIdx.x == 0) {
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < NW; g += 4) s += (red[0][0][g] + red[0][0][g + 1]) + (red[0][0][g + 2] + red[0][0][g + 3]);
      db[co] = accumulate ? db[co] + s : s;
    }
  }
}

struct WgPlan {
  int cb, ib, pw, nco_t, nci_t, npx, npy, npatches, nsplit, per_split;
  size_t slab_floats, bslab_floats;
};

WgPlan make_plan(int N, int H, int W, int Cin_s, int Cout) {
  WgPlan p;
  if (Cout <= 32) { p.cb = 1; p.ib = 4; } else { p.cb = 2; p.ib = 2; }
  p.nco_t = ceil_div(Cout, p.cb * 32);
  p.nci_t = ceil_div(Cin_s, p.ib * 32);
  // 16 x 4 patches when they cover the frame with at least 8 % fewer padded pixels (and the generic 64 x 64 kernel is used)
  p.pw = 32;
  if (p.cb == 2 && (long)ceil_div(W, 16) * 16 * ceil_div(H, 4) * 4 * 100 < (long)ceil_div(W, 32) * 32 * ceil_div(H, 2) * 2 * 92) p.pw = 16;
  {
    OSVOS_ENV_INT(env_pw, "OSVOS_WGRAD_PW", 0);
    if (env_pw > 0 && p.cb == 2) p.pw = env_pw == 16 ? 16 : 32;
  }
  const int PW = p.pw, PH = PPIX / p.pw;
  p.npx = ceil_div(W, PW);
  p.npy = ceil_div(H, PH);
  p.npatches = N * p.npx * p.npy;
  OSVOS_ENV_INT(env_blocks, "OSVOS_WGRAD_BLOCKS", 0);
  // ~2 workgroups per CU; small frames (conv5 at 480p: 30 patches) prefer fewer, longer splits
  int target_blocks = env_blocks > 0 ? env_blocks : (p.npatches >= 100 ? 512 : 256);
  if (target_blocks < 1) target_blocks = 512;
  int want = ceil_div(target_blocks, p.nco_t * p.nci_t);
  int max_split = p.npatches / 4 > 0 ? p.npatches / 4 : 1;
  p.nsplit = want < max_split ? want : max_split;
  if (p.nsplit > 256) p.nsplit = 256;       // (512 splits for the one-tile conv1_2: partial kernel -23 us, slab reduce +37 us)
  if (p.nsplit < 1) p.nsplit = 1;
  p.per_split = ceil_div(p.npatches, p.nsplit);
  p.nsplit = ceil_div(p.npatches, p.per_split);
  p.slab_floats = (size_t)p.nsplit * 9 * Cout * Cin_s;
  p.bslab_floats = (size_t)p.nsplit * Cout;
  return p;
}

template <int CB, int IB, int PIPE, int DBUF, int OCC, int PWT = 32>
int launch_wgrad(const WgArgs& a, long blocks, hipStream_t stream) {
  constexpr size_t lds = (size_t)(DBUF ? 2 : 1) * (PPIX * CB * 32 + Geo<PWT>::XPIX * IB * 32) * 4;
  static bool attr_set_dev[OSVOS_MAX_DEVICES] = {};      // hipFuncSetAttribute is per device
  bool& attr_set = attr_set_dev[osvos_current_device()];
  if (!attr_set) {
    OSVOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_f32_kernel<CB, IB, PIPE, DBUF, OCC, PWT>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL((wgrad_f32_kernel<CB, IB, PIPE, DBUF, OCC, PWT>), dim3((unsigned)blocks), dim3(256), lds, stream, a);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

// tuning knob (tools/tune_conv.py): bit0 PIPE, bit1 DBUF, bit2 OCC=2, bit3 reference-order slabs
int wgrad_variant() {
  OSVOS_ENV_INT(env_variant, "OSVOS_WGRAD_VARIANT", 5);
  return env_variant;     // measured best (tools/tune_wgrad.py): pipelined fetch, one LDS buffer, 2 WGs/CU
}

template <int CB, int IB>
int launch_wgrad_variant(const WgArgs& a, long blocks, hipStream_t stream) {
  switch (wgrad_variant() & 7) {
    case 0: return launch_wgrad<CB, IB, 0, 0, 1>(a, blocks, stream);
    case 1: return launch_wgrad<CB, IB, 1, 0, 1>(a, blocks, stream);
    case 2: return launch_wgrad<CB, IB, 0, 1, 1>(a, blocks, stream);
    case 3: return launch_wgrad<CB, IB, 1, 1, 1>(a, blocks, stream);
    case 4: return launch_wgrad<CB, IB, 0, 0, (CB == 2 ? 2 : 1)>(a, blocks, stream);
    case 5: return launch_wgrad<CB, IB, 1, 0, (CB == 2 ? 2 : 1)>(a, blocks, stream);
    default: return launch_wgrad<CB, IB, 0, 1, 1>(a, blocks, stream);
  }
}

}  // namespace

size_t osvos_wgrad_small_ws_bytes(int N, int H, int W, int Cin_s, int Cout);
int osvos_conv3x3_wgrad_small_f32(const void* x, const void* dy, int wide_bf16, void* ws, float* dw, float* db,
                                  int N, int H, int W, int Cin, int Cin_s, int Cout, int Cout_s,
                                  int accumulate, hipStream_t stream);

int osvos_wgrad_reduce_launch(const float* slab, const float* bslab, float* dw, float* db, int nsplit, int Cout, int Cin,
                              int Cin_s, int accumulate, hipStream_t stream) {
  OSVOS_ENV_INT(env_t, "OSVOS_WGRAD_REDUCE_T", 1);
  if (env_t && Cin == Cin_s && Cin_s % 64 == 0 && Cout <= 65535) {      // wide layers: LDS-transposing reduce, contiguous OIHW writes
    if ((long)(Cin_s / 64) * Cout <= 1024 && nsplit >= 32)      // few workgroups, many splits: 16 waves each
      hipLaunchKernelGGL(wgrad_reduce_t_kernel<16>, dim3(Cin_s / 64, Cout), dim3(1024), 0, stream, slab, bslab, dw, db, nsplit, Cout, Cin_s, accumulate);
    else
      hipLaunchKernelGGL(wgrad_reduce_t_kernel<4>, dim3(Cin_s / 64, Cout), dim3(256), 0, stream, slab, bslab, dw, db, nsplit, Cout, Cin_s, accumulate);
    OSVOS_LAUNCH_CHECK();
    return 0;
  }
  const int total = Cout * Cin_s * 9;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ceil_div(total, 128) + (db ? ceil_div(Cout, 32) : 0)), dim3(256), 0, stream,
                     slab, bslab, dw, db, nsplit, Cout, Cin, Cin_s, accumulate, 0);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

size_t osvos_wgrad_ws_bytes_f32(int N, int H, int W, int Cin_s, int Cout) {
  WgPlan p = make_plan(N, H, W, Cin_s, Cout);
  const size_t generic = align_up((p.slab_floats + p.bslab_floats) * sizeof(float), 256);
  const size_t small = osvos_wgrad_small_ws_bytes(N, H, W, Cin_s, Cout);
  return generic > small ? generic : small;
}

int osvos_conv3x3_wgrad_f32(const float* x, const float* dy, void* ws, float* dw, float* db,
                            int N, int H, int W, int Cin, int Cin_s, int Cout, int Cout_s,
                            int accumulate, hipStream_t stream) {
  OSVOS_ARG_CHECK(x && dy && ws && dw, "wgrad: null pointer");
  OSVOS_ARG_CHECK(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "wgrad: bad shape");
  OSVOS_ARG_CHECK((long)H * W * Cin_s < (1L << 29) && (long)H * W * Cout_s < (1L << 29), "wgrad: image too large for 31-bit byte offsets");
  OSVOS_ARG_CHECK(Cin_s % 4 == 0 && Cout_s % 4 == 0 && Cout % 4 == 0 && Cin <= Cin_s && Cout <= Cout_s,
                  "wgrad f32: channel strides must be multiples of 4 (Cin %d/%d Cout %d/%d)", Cin, Cin_s, Cout, Cout_s);
  {
    OSVOS_ENV_INT(env_generic, "OSVOS_WGRAD_GENERIC", 0);     // tuning / tests: force the generic kernel
    if (!env_generic) {
      const int rc = osvos_conv3x3_wgrad_small_f32(x, dy, 0, ws, dw, db, N, H, W, Cin, Cin_s, Cout, Cout_s, accumulate, stream);
      if (rc <= 0) return rc;
    }
  }
  WgPlan p = make_plan(N, H, W, Cin_s, Cout);
  WgArgs a;
  a.x = x; a.dy = dy;
  a.slab = reinterpret_cast<float*>(ws);
  a.bslab = db ? a.slab + p.slab_floats : nullptr;
  a.N = N; a.H = H; a.W = W; a.Cin_s = Cin_s; a.Cout = Cout; a.Cout_s = Cout_s;
  a.npx = p.npx; a.npy = p.npy; a.npatches = p.npatches; a.nsplit = p.nsplit; a.per_split = p.per_split;
  a.nco_t = p.nco_t; a.nci_t = p.nci_t;
  const long blocks = (long)p.nsplit * p.nco_t * p.nci_t;
  a.oihw = (wgrad_variant() >> 3) & 1;
  const int phase = osvos_wgrad_phase();
  if (phase != 2) {
    int rc = (p.cb == 1) ? launch_wgrad_variant<1, 4>(a, blocks, stream)
                         : (p.pw == 16 ? launch_wgrad<2, 2, 1, 0, 2, 16>(a, blocks, stream)      // (the measured-best variant only)
                                       : launch_wgrad_variant<2, 2>(a, blocks, stream));
    if (rc) return rc;
  }
  if (phase == 1) return 0;
  if (!a.oihw) return osvos_wgrad_reduce_launch(a.slab, a.bslab, dw, db, p.nsplit, Cout, Cin, Cin_s, accumulate, stream);
  const int total = Cout * Cin_s * 9;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ceil_div(total, 128) + (db ? ceil_div(Cout, 32) : 0)), dim3(256), 0, stream,
                     a.slab, a.bslab, dw, db, p.nsplit, Cout, Cin, Cin_s, accumulate, a.oihw);
  OSVOS_LAUNCH_CHECK();
  return 0;
}
