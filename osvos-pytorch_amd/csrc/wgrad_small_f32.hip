// Weight-gradient kernels for the two skinny shapes of the OSVOS graph, where the generic
// 64x64x9-tap kernel (wgrad_f32.hip) would spend most of its MFMAs on zero padding:
//
//   * conv1_1 (Cin = 3, stored NHWC8; reference vgg_osvos.py:36,142): the 27 (tap, ci) pairs become
//     ONE 32-wide MFMA column block ("im2col in the B operand"): D[co][tap*3+ci] += dY[p][co] *
//     X[p+tap][ci].  2 MFMAs per pixel pair instead of 18 -> the kernel is bound by the single
//     read of dY (105 MB at 854x480), not by the matrix pipe.
//   * side_prep (Cout = 16; reference vgg_osvos.py:41): two taps share one 32-row MFMA block,
//     D[(tap&1, co)][ci] += dY[q - shift(tap)][co] * X[q][ci] (the tap shift moves to the dY side, so
//     the X tile needs no halo): 5 MFMAs per pixel pair per 32 cins instead of 9 half-empty ones.
//
// Both write coalesced per-split slabs [split][tap][co][ci] (co16) / [split][co][32] (c3) and share
// the deterministic two-pass reduction idea of the generic kernel.
#include "common.h"

namespace {

constexpr int PW = 32;

// ---------------------------------------------------------------------------------------------
// conv1_1: Cout <= 64 (two 32-row blocks), Cin = 3 in NHWC8
// ---------------------------------------------------------------------------------------------
constexpr int C3_PH = 8, C3_PPIX = PW * C3_PH;                 // 256-pixel patch
constexpr int C3_XW = PW + 2, C3_XH = C3_PH + 2, C3_XPIX = C3_XW * C3_XH;

// 4 consecutive channels at element offset `e`: fp32 tensor, or (bf16-store mode of the network) a bf16 tensor widened on load
__device__ inline f32x4 ld4(const void* base, size_t e, int is_bf16) {
  if (is_bf16) {
    const uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(base) + e);
    return f32x4{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
  }
  return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + e);
}

struct C3Args {
  const float* x;      // NHWC8
  const void* dy;      // NHWC, stride Cout_s; bf16 when dy_bf16
  int dy_bf16;
  float* slab;         // [nsplit][64][32]
  float* bslab;        // [nsplit][64] or null
  int N, H, W, Cout, Cout_s;
  int npx, npy, npatches, per_split;
};

__global__ __launch_bounds__(256, 2) void wgrad_c3_f32_kernel(C3Args a) {
  constexpr int DY_F4 = C3_PPIX * 64 / 4;       // 4096
  constexpr int NDY = DY_F4 / 256;              // 16
  constexpr int NX = (C3_XPIX + 255) / 256;     // 2
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* lds4 = reinterpret_cast<f32x4*>(smem);
  const float* dYs = reinterpret_cast<const float*>(smem);
  const float* Xs = dYs + DY_F4 * 4;            // [C3_XPIX][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int split = blockIdx.x;
  const int p_begin = split * a.per_split, p_end = min(p_begin + a.per_split, a.npatches);

  // B-operand column j = tap*3 + ci -> float offset inside the X halo tile; columns 27..31 read the
  // zero pad channel (channel 3 of NHWC8 is zero filled)
  const int tap = li / 3, ci = li % 3;
  const int boff = li < 27 ? ((tap / 3) * C3_XW + (tap % 3)) * 4 + ci : 3;

  f32x4 rdy[NDY], rx[NX];
  auto load_patch = [&](int p) {
    const int px = p % a.npx;
    int t = p / a.npx;
    const int py = t % a.npy;
    const int n = t / a.npy;
    const int x0 = px * PW, y0 = py * C3_PH;
#pragma unroll
    for (int i = 0; i < NDY; ++i) {
      const int e = tid + i * 256;
      const int pix = e >> 4, q = e & 15;
      const int gy = y0 + pix / PW, gx = x0 + pix % PW, co = 4 * q;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (gy < a.H && gx < a.W && co < a.Cout)
        v = ld4(a.dy, ((size_t)(n * a.H + gy) * a.W + gx) * a.Cout_s + co, a.dy_bf16);
      rdy[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int e = tid + i * 256;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (e < C3_XPIX) {
        const int gy = y0 + e / C3_XW - 1, gx = x0 + e % C3_XW - 1;
        if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
          v = *reinterpret_cast<const f32x4*>(a.x + ((size_t)(n * a.H + gy) * a.W + gx) * 8);
      }
      rx[i] = v;
    }
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int i = 0; i < NDY; ++i) lds4[tid + i * 256] = rdy[i];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int e = tid + i * 256;
      if (e < C3_XPIX) lds4[DY_F4 + e] = rx[i];
    }
  };

  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  float bsum = 0.f;
  const int bco = tid & 63, bgrp = tid >> 6;

  if (p_begin < p_end) load_patch(p_begin);
  for (int p = p_begin; p < p_end; ++p) {
    __syncthreads();
    store_patch();
    __syncthreads();
    if (p + 1 < p_end) load_patch(p + 1);
    // wave w multiplies patch rows 2w, 2w+1 (32 pixel pairs)
#pragma unroll 4
    for (int pp = 0; pp < 32; ++pp) {
      const int dy = 2 * wave + pp / 16, dx = (pp % 16) * 2 + lh;
      const float* arow = dYs + (dy * PW + dx) * 64 + li;
      const float bv = Xs[(dy * C3_XW + dx) * 4 + boff];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[0], bv, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[32], bv, acc1, 0, 0, 0);
    }
    if (a.bslab != nullptr) {
#pragma unroll 8
      for (int pix = bgrp; pix < C3_PPIX; pix += 4) bsum += dYs[pix * 64 + bco];
    }
  }
  __syncthreads();
  // combine the four waves' partial tiles through LDS, then one coalesced slab write
  float* red = reinterpret_cast<float*>(smem);      // [4 waves][2 blocks][16 regs][64 lanes]
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    red[((wave * 2 + 0) * 16 + r) * 64 + lane] = acc0[r];
    red[((wave * 2 + 1) * 16 + r) * 64 + lane] = acc1[r];
  }
  float* bred = red + 4 * 2 * 16 * 64;
  bred[tid] = bsum;
  __syncthreads();
  for (int e = tid; e < 2 * 16 * 64; e += 256) {
    const float s = red[e] + red[2048 + e] + red[4096 + e] + red[6144 + e];
    const int ln = e & 63, r = (e >> 6) & 15, blk = e >> 10;
    const int co = blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5), j = ln & 31;
    a.slab[((size_t)split * 64 + co) * 32 + j] = s;
  }
  if (a.bslab != nullptr && tid < 64)
    a.bslab[(size_t)split * 64 + tid] = bred[tid] + bred[64 + tid] + bred[128 + tid] + bred[192 + tid];
}

// ---- conv1_1 weight gradient of the bf16-store mode on the bf16 matrix pipe ------------------------------------------------------------
// dy arrives as bf16 (the trunk tensors of that mode); x (the 3-channel input, NHWC8 fp32) is rounded to bf16 like the mode's forward conv1_1
// does while staging.  The fp32 kernel above walks 256-pixel patches with ONE workgroup per CU, dY widened to fp32 in LDS (64 KB) and two
// fp32 MFMAs per pixel pair: 1.12 ms per step at batch 12 (profiles/r02_bench_bf16_b12_kernel_stats.txt) against ~0.15 ms that reading the
// 630 MB of dY takes.  Here: 16 x 8 pixel patches, the dY tile stays bf16 and pixel-major (24 KB), the 27 (tap, ci) columns are built ONCE
// per patch as a [pixel][32] bf16 im2col tile (8 KB) from the fp32 halo, both operands are gathered with ds_read_b64_tr_b16 and a patch
// costs each wave 4 MFMAs (32x32x16); ~36 KB of LDS -> four workgroups per CU, next patch's loads in flight during the current one.
// Same slab layout [split][64 co][32 j] and reduce kernel as above; bias gradient = fp32 column sums of the bf16 dY.
constexpr int Q_W = 16, Q_H = 8, Q_PIX = Q_W * Q_H, Q_HW = Q_W + 2, Q_HH = Q_H + 2, Q_HPIX = Q_HW * Q_HH;
constexpr int Q_DYP = 64 * 2 + 64, Q_BP = 64;                    // byte pitches of the dY tile (64 ch + skew) and the im2col tile (32 cols)
constexpr int Q_DY_B = Q_PIX * Q_DYP, Q_XH_B = Q_HPIX * 16, Q_B_B = Q_PIX * Q_BP;
constexpr size_t Q_LDS = (size_t)Q_DY_B + Q_XH_B + Q_B_B;

__global__ __launch_bounds__(256, 4) void wgrad_c3_bf16_kernel(C3Args a) {
  typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* dYs = smem;
  float* Xh = reinterpret_cast<float*>(smem + Q_DY_B);          // [Q_HPIX][4]
  char* Bs = smem + Q_DY_B + Q_XH_B;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int split = blockIdx.x;
  const int p_begin = split * a.per_split, p_end = min(p_begin + a.per_split, a.npatches);
  constexpr unsigned OOB = 0x80000000u;
  const int oct = tid & 7, pg = tid >> 3;                        // dY items: (pixel pg + 32 i, channel octet)
  const bool ch_ok = 8 * oct < a.Cout;
  const int img_dy_bytes = a.H * a.W * a.Cout_s * 2, img_x_bytes = a.H * a.W * 8 * 4;

  u32x4 rdy[4], rx;
  struct Patch { __amdgpu_buffer_rsrc_t drs, xrs; int x0, y0; };
  auto locate = [&](int p, bool live) -> Patch {
    const int px = p % a.npx;
    int t = p / a.npx;
    const int py = t % a.npy;
    const int n = live ? t / a.npy : 0;
    Patch q;
    q.x0 = live ? px * Q_W : 0x40000000;
    q.y0 = py * Q_H;
    q.drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(a.dy)) + (size_t)n * img_dy_bytes, 0, img_dy_bytes, 0x00020000);
    q.xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(a.x)) + (size_t)n * img_x_bytes, 0, img_x_bytes, 0x00020000);
    return q;
  };
  auto load_patch = [&](const Patch& q) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = pg + 32 * i, py = p / Q_W, pxx = p % Q_W;
      const unsigned off = (ch_ok && q.x0 + pxx < a.W) ? (unsigned)((((q.y0 + py) * a.W + q.x0 + pxx) * a.Cout_s + 8 * oct) * 2) : OOB;
      rdy[i] = __builtin_amdgcn_raw_buffer_load_b128(q.drs, off, 0, 0);
    }
    const int hy = tid / Q_HW, hx = tid % Q_HW;
    const bool ok = tid < Q_HPIX && (unsigned)(q.x0 - 1 + hx) < (unsigned)a.W;
    const unsigned off = ok ? (unsigned)((((q.y0 - 1 + hy) * a.W + q.x0 - 1 + hx) * 8) * 4) : OOB;      // rows outside fall out of the buffer range
    rx = __builtin_amdgcn_raw_buffer_load_b128(q.xrs, off, 0, 0);
  };

  // im2col column j = tap * 3 + ci -> float offset inside the halo tile relative to the pixel; columns 27..31 read channel 3 (zero pad of NHWC8)
  const int bpx = tid >> 1, bhalf = tid & 1;
  const int bpy = bpx / Q_W, bpxx = bpx % Q_W;
  const int fi = lane & 15, fg = (lane >> 4) & 1, lh = lane >> 5;
  const int cb = wave & 1, kh = wave >> 1;                       // cout block, patch-row half
  const char* a_base = dYs + (8 * lh + (fi >> 2)) * Q_DYP + (32 * cb + 16 * fg + 4 * (fi & 3)) * 2;
  const char* b_base = Bs + (8 * lh + (fi >> 2)) * Q_BP + (16 * fg + 4 * (fi & 3)) * 2;
  auto tr8 = [&](const char* p, int pitch) -> s16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * pitch));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool want_bias = a.bslab != nullptr;

  load_patch(locate(p_begin, p_begin < p_end));
  for (int p = p_begin; p < p_end; ++p) {
    __syncthreads();                       // every wave is done with the previous patch's tiles
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<u32x4*>(dYs + (pg + 32 * i) * Q_DYP + oct * 16) = rdy[i];
      if (want_bias) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          bsum[2 * d] += __uint_as_float(rdy[i][d] << 16);
          bsum[2 * d + 1] += __uint_as_float(rdy[i][d] & 0xffff0000u);
        }
      }
    }
    if (tid < Q_HPIX) *reinterpret_cast<u32x4*>(Xh + tid * 4) = rx;
    __syncthreads();
    load_patch(locate(p + 1, p + 1 < p_end));      // in flight during the rest of this patch
    {   // build this thread's 16 im2col columns of its pixel
      unsigned w[8];
#pragma unroll
      for (int jj = 0; jj < 16; jj += 2) {
        float v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int j = 16 * bhalf + jj + u;
          const int tap = j / 3, ci = j - 3 * tap;
          const int off = j < 27 ? (((bpy + tap / 3) * Q_HW + bpxx + tap % 3) * 4 + ci) : 3;
          v[u] = Xh[off];
        }
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        bf16x2_t h;
        h[0] = (__bf16)v[0];
        h[1] = (__bf16)v[1];
        w[jj >> 1] = __builtin_bit_cast(unsigned, h);
      }
      *reinterpret_cast<u32x4*>(Bs + bpx * Q_BP + bhalf * 32) = u32x4{w[0], w[1], w[2], w[3]};
      *reinterpret_cast<u32x4*>(Bs + bpx * Q_BP + bhalf * 32 + 16) = u32x4{w[4], w[5], w[6], w[7]};
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int row = 4 * kh + ks;
      const s16x8 fa = tr8(a_base + row * Q_W * Q_DYP, Q_DYP);
      const s16x8 fb = tr8(b_base + row * Q_W * Q_BP, Q_BP);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb), __builtin_bit_cast(bf16x8_t, fa), acc, 0, 0, 0);
    }
  }
  __syncthreads();
  // D = [j rows][co columns]: lane (li, lh) holds co li of its block and columns j = 8 q + 4 lh + (0..3) in registers 4q..4q+3
  float* red = reinterpret_cast<float*>(smem);      // [4 waves][16 regs][64 lanes]
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
  float* bred = red + 4 * 16 * 64;                  // [32 pixel groups][64 channels]
  if (want_bias)
#pragma unroll
    for (int c = 0; c < 8; ++c) bred[pg * 64 + oct * 8 + c] = bsum[c];
  __syncthreads();
  for (int e = tid; e < 2 * 16 * 64; e += 256) {
    const int ln = e & 63, r = (e >> 6) & 15, blk = e >> 10;      // blk = cout block; its two row-half waves are blk and blk + 2
    const float s = red[(blk * 16 + r) * 64 + ln] + red[((blk + 2) * 16 + r) * 64 + ln];
    const int co = blk * 32 + (ln & 31), j = (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
    a.slab[((size_t)split * 64 + co) * 32 + j] = s;
  }
  if (want_bias && tid < 64) {
    float s = 0.f;
    for (int g = 0; g < 32; ++g) s += bred[g * 64 + tid];
    a.bslab[(size_t)split * 64 + tid] = s;
  }
}

// one wave per 4 outputs: lanes stride over the splits, shuffle-reduce (405 splits x 1728 outputs would
// otherwise be 405 sequential loads per thread)
__global__ __launch_bounds__(256) void wgrad_c3_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bslab,
                                                              float* __restrict__ dw, float* __restrict__ db, int nsplit, int Cout,
                                                              int accumulate) {
  const int lane = threadIdx.x & 63;
  const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);      // co*27 + ci*9 + tap (OIHW, Cin = 3); then Cout bias sums
  if (idx < Cout * 27) {
    const int co = idx / 27, rem = idx % 27, ci = rem / 9, tap = rem % 9;
    float s = 0.f;
    for (int sp = lane; sp < nsplit; sp += 64) s += slab[((size_t)sp * 64 + co) * 32 + tap * 3 + ci];
    s = wave_sum(s);
    if (lane == 0) dw[idx] = accumulate ? dw[idx] + s : s;
  } else if (db != nullptr && idx < Cout * 27 + Cout) {
    const int co = idx - Cout * 27;
    float s = 0.f;
    for (int sp = lane; sp < nsplit; sp += 64) s += bslab[(size_t)sp * 64 + co];
    s = wave_sum(s);
    if (lane == 0) db[co] = accumulate ? db[co] + s : s;
  }
}

// ---------------------------------------------------------------------------------------------
// side_prep: Cout = 16, Cin a multiple of 32
// ---------------------------------------------------------------------------------------------
constexpr int S_PH = 2, S_PPIX = PW * S_PH;                    // 64-pixel patch
constexpr int S_YW = PW + 2, S_YH = S_PH + 2, S_YPIX = S_YW * S_YH;   // dY halo
constexpr int S_BCI = 128;                                     // cins per workgroup (one 32-block per wave)

struct S16Args {
  const void* x;       // NHWC stride Cin_s; bf16 when x_bf16
  int x_bf16;
  const float* dy;     // NHWC stride Cout_s, 16 channels used
  float* slab;         // [nsplit][9][16][Cin_s]
  float* bslab;        // [nsplit][16] or null
  int N, H, W, Cin_s, Cout_s;
  int npx, npy, npatches, per_split, nci_t;
};

__global__ __launch_bounds__(256, 2) void wgrad_co16_f32_kernel(S16Args a) {
  constexpr int X_F4 = S_PPIX * S_BCI / 4;      // 2048
  constexpr int NX = X_F4 / 256;                // 8
  constexpr int Y_F4 = S_YPIX * 16 / 4;         // 544
  constexpr int NY = (Y_F4 + 255) / 256;        // 3
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* lds4 = reinterpret_cast<f32x4*>(smem);
  const float* Xs = reinterpret_cast<const float*>(smem);      // [64 px][128]
  const float* Ys = Xs + X_F4 * 4;                             // [136 halo px][16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int cit = blockIdx.x % a.nci_t, split = blockIdx.x / a.nci_t;
  const int ci0 = cit * S_BCI;
  const int p_begin = split * a.per_split, p_end = min(p_begin + a.per_split, a.npatches);

  // A-operand row i = (slot, co): tap = 2g + slot reads dY at halo (dy + 2 - r, dx + 2 - s)
  const int slot = li >> 4, co = li & 15;
  int aoff[5];
  float amul[5];
#pragma unroll
  for (int g = 0; g < 5; ++g) {
    const int t = 2 * g + slot;
    const int r = t < 9 ? t / 3 : 0, s = t < 9 ? t % 3 : 0;
    aoff[g] = ((2 - r) * S_YW + (2 - s)) * 16 + co;
    amul[g] = t < 9 ? 1.f : 0.f;
  }

  f32x4 rx[NX], ry[NY];
  auto load_patch = [&](int p) {
    const int px = p % a.npx;
    int t = p / a.npx;
    const int py = t % a.npy;
    const int n = t / a.npy;
    const int x0 = px * PW, y0 = py * S_PH;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int e = tid + i * 256;
      const int pix = e / (S_BCI / 4), q = e % (S_BCI / 4);
      const int gy = y0 + pix / PW, gx = x0 + pix % PW, ci = ci0 + 4 * q;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (gy < a.H && gx < a.W && ci < a.Cin_s)
        v = ld4(a.x, ((size_t)(n * a.H + gy) * a.W + gx) * a.Cin_s + ci, a.x_bf16);
      rx[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      const int e = tid + i * 256;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (e < Y_F4) {
        const int pix = e >> 2, q = e & 3;
        const int gy = y0 + pix / S_YW - 1, gx = x0 + pix % S_YW - 1;
        if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
          v = *reinterpret_cast<const f32x4*>(a.dy + ((size_t)(n * a.H + gy) * a.W + gx) * a.Cout_s + 4 * q);
      }
      ry[i] = v;
    }
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int i = 0; i < NX; ++i) lds4[tid + i * 256] = rx[i];
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      const int e = tid + i * 256;
      if (e < Y_F4) lds4[X_F4 + e] = ry[i];
    }
  };

  f32x16 acc[5];
#pragma unroll
  for (int g = 0; g < 5; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
  float bsum = 0.f;
  const bool do_bias = a.bslab != nullptr && cit == 0;

  if (p_begin < p_end) load_patch(p_begin);
  for (int p = p_begin; p < p_end; ++p) {
    __syncthreads();
    store_patch();
    __syncthreads();
    if (p + 1 < p_end) load_patch(p + 1);
#pragma unroll 2
    for (int pp = 0; pp < S_PPIX / 2; ++pp) {
      const int dy = pp / (PW / 2), dx = (pp % (PW / 2)) * 2 + lh;
      const float bv = Xs[(dy * PW + dx) * S_BCI + wave * 32 + li];
      const float* yb = Ys + (dy * S_YW + dx) * 16;
#pragma unroll
      for (int g = 0; g < 5; ++g)
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(yb[aoff[g]] * amul[g], bv, acc[g], 0, 0, 0);
    }
    if (do_bias && tid < 16) {       // column sums of the patch's own (non-halo) dY pixels
      for (int pix = 0; pix < S_PPIX; ++pix) bsum += Ys[((pix / PW + 1) * S_YW + pix % PW + 1) * 16 + tid];
    }
  }
  // D row i = (slot, co) = (r&3) + 8*(r>>2) + 4*lh, col = cin li of this wave's 32-block
  const int ci = ci0 + wave * 32 + li;
#pragma unroll
  for (int g = 0; g < 5; ++g) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int t = 2 * g + (i >> 4);
      if (t < 9 && ci < a.Cin_s) a.slab[((size_t)(split * 9 + t) * 16 + (i & 15)) * a.Cin_s + ci] = acc[g][r];
    }
  }
  if (do_bias && tid < 16) a.bslab[(size_t)split * 16 + tid] = bsum;
}

struct SmallPlan {
  int npx, npy, npatches, nsplit, per_split, nci_t;
  size_t slab_floats, bslab_floats;
};

SmallPlan plan_c3(int N, int H, int W) {
  SmallPlan p;
  p.npx = ceil_div(W, PW); p.npy = ceil_div(H, C3_PH); p.npatches = N * p.npx * p.npy; p.nci_t = 1;
  int want = 256;    // runs on the aux stream next to the dgrad chain: fewer, longer splits, smaller slabs
  if (want > p.npatches) want = p.npatches;
  p.per_split = ceil_div(p.npatches, want);
  p.nsplit = ceil_div(p.npatches, p.per_split);
  p.slab_floats = (size_t)p.nsplit * 64 * 32;
  p.bslab_floats = (size_t)p.nsplit * 64;
  return p;
}

// bf16-operand form: 16 x 8 pixel patches, four workgroups per CU
SmallPlan plan_c3_bf16(int N, int H, int W) {
  SmallPlan p;
  p.npx = ceil_div(W, Q_W); p.npy = ceil_div(H, Q_H); p.npatches = N * p.npx * p.npy; p.nci_t = 1;
  int want = 1024;
  if (want > p.npatches) want = p.npatches;
  p.per_split = ceil_div(p.npatches, want);
  p.nsplit = ceil_div(p.npatches, p.per_split);
  p.slab_floats = (size_t)p.nsplit * 64 * 32;
  p.bslab_floats = (size_t)p.nsplit * 64;
  return p;
}

SmallPlan plan_co16(int N, int H, int W, int Cin_s) {
  SmallPlan p;
  p.npx = ceil_div(W, PW); p.npy = ceil_div(H, S_PH); p.npatches = N * p.npx * p.npy;
  p.nci_t = ceil_div(Cin_s, S_BCI);
  int want = ceil_div(256, p.nci_t);
  const int max_split = p.npatches / 4 > 0 ? p.npatches / 4 : 1;
  if (want > max_split) want = max_split;
  p.per_split = ceil_div(p.npatches, want);
  p.nsplit = ceil_div(p.npatches, p.per_split);
  p.slab_floats = (size_t)p.nsplit * 9 * 16 * Cin_s;
  p.bslab_floats = (size_t)p.nsplit * 16;
  return p;
}

int g_c3_bf16 = 1;      // the bf16-operand conv1_1 weight gradient (osvos_debug_set_c3_bf16: tests compare it with the fp32 kernel)

}  // namespace

extern "C" int osvos_debug_set_c3_bf16(int on) {
  const int prev = g_c3_bf16;
  g_c3_bf16 = on ? 1 : 0;
  return prev;
}

// generic slab reduce of wgrad_f32.hip (layout [split][tap][co][ci])
int osvos_wgrad_reduce_launch(const float* slab, const float* bslab, float* dw, float* db, int nsplit, int Cout, int Cin,
                              int Cin_s, int accumulate, hipStream_t stream);

size_t osvos_wgrad_small_ws_bytes(int N, int H, int W, int Cin_s, int Cout) {
  if (Cin_s == 8 && Cout <= 64) {
    SmallPlan p = plan_c3(N, H, W), q = plan_c3_bf16(N, H, W);
    const size_t a = p.slab_floats + p.bslab_floats, b = q.slab_floats + q.bslab_floats;
    return align_up((a > b ? a : b) * sizeof(float), 256);
  }
  if (Cout == 16) {
    SmallPlan p = plan_co16(N, H, W, Cin_s);
    return align_up((p.slab_floats + p.bslab_floats) * sizeof(float), 256);
  }
  return 0;
}

// returns 1 if the shape is not one of the two special cases (caller falls through to the generic kernel)
// wide_bf16: the WIDE operand is a bf16 tensor (dy of the Cin = 3 layer, x of the Cout = 16 layers); the narrow one stays fp32
int osvos_conv3x3_wgrad_small_f32(const void* x, const void* dy, int wide_bf16, void* ws, float* dw, float* db,
                                  int N, int H, int W, int Cin, int Cin_s, int Cout, int Cout_s,
                                  int accumulate, hipStream_t stream) {
  if (Cin == 3 && Cin_s == 8 && Cout == 64 && Cout_s % 8 == 0 && wide_bf16 && g_c3_bf16) {      // bf16-store mode: dY is bf16 -> bf16 matrix pipe
    SmallPlan p = plan_c3_bf16(N, H, W);
    C3Args a;
    a.x = reinterpret_cast<const float*>(x); a.dy = dy; a.dy_bf16 = 1;
    a.slab = reinterpret_cast<float*>(ws);
    a.bslab = db ? a.slab + p.slab_floats : nullptr;
    a.N = N; a.H = H; a.W = W; a.Cout = Cout; a.Cout_s = Cout_s;
    a.npx = p.npx; a.npy = p.npy; a.npatches = p.npatches; a.per_split = p.per_split;
    OSVOS_ARG_CHECK((long)H * W * Cout_s < (1L << 30), "wgrad c3 bf16: image too large for 31-bit byte offsets");
    const int phase = osvos_wgrad_phase();
    if (phase != 2) {
      hipLaunchKernelGGL(wgrad_c3_bf16_kernel, dim3(p.nsplit), dim3(256), Q_LDS, stream, a);
      OSVOS_LAUNCH_CHECK();
    }
    if (phase == 1) return 0;
    hipLaunchKernelGGL(wgrad_c3_reduce_kernel, dim3(ceil_div(Cout * 28, 4)), dim3(256), 0, stream,
                       a.slab, a.bslab, dw, db, p.nsplit, Cout, accumulate);
    OSVOS_LAUNCH_CHECK();
    return 0;
  }
  if (Cin == 3 && Cin_s == 8 && Cout <= 64 && Cout % 4 == 0) {
    SmallPlan p = plan_c3(N, H, W);
    C3Args a;
    a.x = reinterpret_cast<const float*>(x); a.dy = dy; a.dy_bf16 = wide_bf16;
    a.slab = reinterpret_cast<float*>(ws);
    a.bslab = db ? a.slab + p.slab_floats : nullptr;
    a.N = N; a.H = H; a.W = W; a.Cout = Cout; a.Cout_s = Cout_s;
    a.npx = p.npx; a.npy = p.npy; a.npatches = p.npatches; a.per_split = p.per_split;
    constexpr size_t lds = (size_t)(C3_PPIX * 64 + C3_XPIX * 4) * 4;
    static bool attr_set_dev[OSVOS_MAX_DEVICES] = {};      // hipFuncSetAttribute is per device
  bool& attr_set = attr_set_dev[osvos_current_device()];
    if (!attr_set) {
      OSVOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_c3_f32_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_set = true;
    }
    const int phase = osvos_wgrad_phase();
    if (phase != 2) {
      hipLaunchKernelGGL(wgrad_c3_f32_kernel, dim3(p.nsplit), dim3(256), lds, stream, a);
      OSVOS_LAUNCH_CHECK();
    }
    if (phase == 1) return 0;
    hipLaunchKernelGGL(wgrad_c3_reduce_kernel, dim3(ceil_div(Cout * 28, 4)), dim3(256), 0, stream,
                       a.slab, a.bslab, dw, db, p.nsplit, Cout, accumulate);
    OSVOS_LAUNCH_CHECK();
    return 0;
  }
  if (Cout == 16 && Cout_s % 4 == 0 && Cin_s % 32 == 0 && Cin == Cin_s) {
    SmallPlan p = plan_co16(N, H, W, Cin_s);
    S16Args a;
    a.x = x; a.x_bf16 = wide_bf16; a.dy = reinterpret_cast<const float*>(dy);
    a.slab = reinterpret_cast<float*>(ws);
    a.bslab = db ? a.slab + p.slab_floats : nullptr;
    a.N = N; a.H = H; a.W = W; a.Cin_s = Cin_s; a.Cout_s = Cout_s;
    a.npx = p.npx; a.npy = p.npy; a.npatches = p.npatches; a.per_split = p.per_split; a.nci_t = p.nci_t;
    constexpr size_t lds = (size_t)(S_PPIX * S_BCI + S_YPIX * 16) * 4;
    const int phase = osvos_wgrad_phase();
    if (phase != 2) {
      hipLaunchKernelGGL(wgrad_co16_f32_kernel, dim3(p.nsplit * p.nci_t), dim3(256), lds, stream, a);
      OSVOS_LAUNCH_CHECK();
    }
    if (phase == 1) return 0;
    return osvos_wgrad_reduce_launch(a.slab, a.bslab, dw, db, p.nsplit, 16, Cin, Cin_s, accumulate, stream);
  }
  return 1;
}
