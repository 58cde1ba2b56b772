// 2x2 stride-2 max-pool with ceil_mode=True, NHWC (reference vgg_osvos.py:140,
// nn.MaxPool2d(kernel_size=2, stride=2, ceil_mode=True) -> aten::max_pool2d_with_indices) and its
// backward fused with the ReLU backward of the producing conv and the side-branch gradient add.
// HBM-bound glue, no indices stored -- the backward recomputes the argmax from the saved (post-ReLU) pool input.
//
// Work item = one 2x2 window x one 16-byte channel group (4 fp32 / 8 bf16 channels).  A workgroup owns a run of consecutive items of
// ONE output row (block -> (image, output row, segment): three block-uniform divisions, everything per lane is 32-bit shifts and adds),
// so a wave's loads are the contiguous even / odd pixels of two input rows and every load of a thread's items is in flight before the
// first one is used.  (The first form flattened (n, oy, ox, c) into one 64-bit index and divided it three times per lane: ~600 VALU
// instructions per 208 bytes moved, a third of the kernel's time at 2.5 TB/s -- profiles/r03_ab_glue_kernels.txt.)
#include "common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int NT = 256;

struct PoolGeo {
  int H, W, CG, Ho, Wo;      // CG: 16-byte channel groups per pixel
  int segs, shift;           // segments per output row; log2(CG) (-1: not a power of two)
  unsigned row_items;        // Wo * CG
};

// element access inside a 16-byte group
struct F32E {
  static constexpr int K = 4;
  static __device__ inline float get(const u32x4& v, int k) { return __uint_as_float(v[k]); }
  static __device__ inline void put(u32x4& v, int k, float f) { v[k] = __float_as_uint(f); }
};
struct BF16E {
  static constexpr int K = 8;
  static __device__ inline float get(const u32x4& v, int k) { return __uint_as_float((k & 1) ? (v[k >> 1] & 0xffff0000u) : (v[k >> 1] << 16)); }
  static __device__ inline void put(u32x4& v, int k, float f) {      // round to nearest even, like every other bf16 store of the library
    const unsigned b = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)f);
    v[k >> 1] = (k & 1) ? (v[k >> 1] | (b << 16)) : b;
  }
};

typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
__device__ inline uint2 to_bf16x4(const u32x4& v) {
  bf16x4_t h;
  h[0] = (__bf16)__uint_as_float(v[0]); h[1] = (__bf16)__uint_as_float(v[1]); h[2] = (__bf16)__uint_as_float(v[2]); h[3] = (__bf16)__uint_as_float(v[3]);
  return __builtin_bit_cast(uint2, h);
}

// block -> (n, oy, first item of its segment)
__device__ inline void block_coords(const PoolGeo& g, unsigned& n, unsigned& oy, unsigned& j0, int items_per_block) {
  unsigned b = blockIdx.x;
  const unsigned seg = b % (unsigned)g.segs;
  b /= (unsigned)g.segs;
  oy = b % (unsigned)g.Ho;
  n = b / (unsigned)g.Ho;
  j0 = seg * (unsigned)items_per_block;
}
__device__ inline void item_coords(const PoolGeo& g, unsigned j, unsigned& ox, unsigned& c) {
  ox = g.shift >= 0 ? (j >> g.shift) : (j / (unsigned)g.CG);
  c = j - ox * (unsigned)g.CG;
}

// ybf (optional, fp32 only): bf16 copy of the result, the operand format of the bf16-MFMA convolutions that consume it
template <class E, int IT>
__global__ __launch_bounds__(NT) void maxpool_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ y, uint2* __restrict__ ybf, PoolGeo g) {
  unsigned n, oy, j0;
  block_coords(g, n, oy, j0, NT * IT);
  const unsigned iy = 2 * oy;
  const bool vy = iy + 1 < (unsigned)g.H;      // clipped (never padded) partial windows
  const u32x4* r0 = x + ((size_t)n * g.H + iy) * g.W * g.CG;
  const u32x4* r1 = r0 + (size_t)g.W * g.CG;
  u32x4 v[IT][4];
  bool live[IT], vx[IT];
  unsigned j[IT];
#pragma unroll
  for (int t = 0; t < IT; ++t) {
    j[t] = j0 + t * NT + threadIdx.x;
    live[t] = j[t] < g.row_items;
    unsigned ox, c;
    item_coords(g, j[t], ox, c);
    vx[t] = 2 * ox + 1 < (unsigned)g.W;
    const unsigned o = 2 * ox * g.CG + c;
    if (live[t]) {
      v[t][0] = r0[o];
      if (vx[t]) v[t][1] = r0[o + g.CG];
      if (vy) v[t][2] = r1[o];
      if (vy && vx[t]) v[t][3] = r1[o + g.CG];
    }
  }
  u32x4* yr = y + ((size_t)n * g.Ho + oy) * g.row_items;
#pragma unroll
  for (int t = 0; t < IT; ++t) {
    if (!live[t]) continue;
    const bool valid[4] = {true, vx[t], vy, vx[t] && vy};
    u32x4 m = u32x4{0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < E::K; ++k) {
      float best = E::get(v[t][0], k);
#pragma unroll
      for (int q = 1; q < 4; ++q)
        if (valid[q] && E::get(v[t][q], k) > best) best = E::get(v[t][q], k);
      E::put(m, k, best);      // exact: the winner is an element of the input's own type
    }
    yr[j[t]] = m;
    if (E::K == 4 && ybf != nullptr) ybf[((size_t)n * g.Ho + oy) * g.row_items + j[t]] = to_bf16x4(m);
  }
}

// dx[pos] = (x[pos] > 0) * ( (pos == first argmax of the window) * dy + dside[pos] )      (bf16: arithmetic in fp32, result rounded RNE)
template <class E, int IT>
__global__ __launch_bounds__(NT) void maxpool_bwd_kernel(const u32x4* __restrict__ x, const u32x4* __restrict__ dy, const u32x4* __restrict__ dside,
                                                         u32x4* __restrict__ dx, uint2* __restrict__ dxbf, PoolGeo g) {
  unsigned n, oy, j0;
  block_coords(g, n, oy, j0, NT * IT);
  const unsigned iy = 2 * oy;
  const bool vy = iy + 1 < (unsigned)g.H;
  const size_t row0 = ((size_t)n * g.H + iy) * g.W * g.CG, row1 = row0 + (size_t)g.W * g.CG;
  const u32x4* gr = dy + ((size_t)n * g.Ho + oy) * g.row_items;
  const bool side = dside != nullptr;
  u32x4 v[IT][4], s[IT][4], gq[IT];
  bool live[IT], vx[IT];
  unsigned o[IT];
#pragma unroll
  for (int t = 0; t < IT; ++t) {
    const unsigned j = j0 + t * NT + threadIdx.x;
    live[t] = j < g.row_items;
    unsigned ox, c;
    item_coords(g, j, ox, c);
    vx[t] = 2 * ox + 1 < (unsigned)g.W;
    o[t] = 2 * ox * g.CG + c;
#pragma unroll
    for (int q = 0; q < 4; ++q) { v[t][q] = u32x4{0, 0, 0, 0}; s[t][q] = u32x4{0, 0, 0, 0}; }
    if (live[t]) {
      gq[t] = gr[j];
      v[t][0] = x[row0 + o[t]];
      if (vx[t]) v[t][1] = x[row0 + o[t] + g.CG];
      if (vy) v[t][2] = x[row1 + o[t]];
      if (vy && vx[t]) v[t][3] = x[row1 + o[t] + g.CG];
      if (side) {
        s[t][0] = dside[row0 + o[t]];
        if (vx[t]) s[t][1] = dside[row0 + o[t] + g.CG];
        if (vy) s[t][2] = dside[row1 + o[t]];
        if (vy && vx[t]) s[t][3] = dside[row1 + o[t] + g.CG];
      }
    }
  }
#pragma unroll
  for (int t = 0; t < IT; ++t) {
    if (!live[t]) continue;
    const bool valid[4] = {true, vx[t], vy, vx[t] && vy};
    u32x4 out[4] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}};
#pragma unroll
    for (int k = 0; k < E::K; ++k) {
      int bi = 0;
      float best = E::get(v[t][0], k);
#pragma unroll
      for (int q = 1; q < 4; ++q)          // scan order (0,0) (0,1) (1,0) (1,1); strict > keeps the first max
        if (valid[q] && E::get(v[t][q], k) > best) { best = E::get(v[t][q], k); bi = q; }
      const float gk = E::get(gq[t], k);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float r = (q == bi ? gk : 0.f) + E::get(s[t][q], k);
        E::put(out[q], k, E::get(v[t][q], k) > 0.f ? r : 0.f);
      }
    }
    const size_t at[4] = {row0 + o[t], row0 + o[t] + g.CG, row1 + o[t], row1 + o[t] + g.CG};
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (valid[q]) {
        dx[at[q]] = out[q];
        if (E::K == 4 && dxbf != nullptr) dxbf[at[q]] = to_bf16x4(out[q]);
      }
  }
}

inline int make_geo(PoolGeo& g, int N, int H, int W, int CG, int items_per_block, long& blocks) {
  g.H = H; g.W = W; g.CG = CG;
  g.Ho = (H + 1) / 2; g.Wo = (W + 1) / 2;
  g.row_items = (unsigned)g.Wo * (unsigned)CG;
  g.segs = ceil_div((int)g.row_items, items_per_block);
  g.shift = -1;
  for (int s = 0; s < 16; ++s)
    if ((1 << s) == CG) g.shift = s;
  blocks = (long)N * g.Ho * g.segs;
  // per-lane offsets inside one input row pair are 32-bit item counts; the row bases are 64-bit
  return ((long)W * CG * 2 < (1L << 31) && blocks < (1L << 31)) ? 0 : 1;
}

constexpr int IT_F32 = 2, IT_BF16 = 1;

}  // namespace

int osvos_maxpool2x2_f32(const float* x, float* y, void* ybf, int N, int H, int W, int C, hipStream_t stream) {
  OSVOS_ARG_CHECK(x && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "maxpool: bad arguments (C=%d)", C);
  PoolGeo g;
  long blocks;
  OSVOS_ARG_CHECK(make_geo(g, N, H, W, C / 4, NT * IT_F32, blocks) == 0, "maxpool: tensor too large (%dx%dx%dx%d)", N, H, W, C);
  hipLaunchKernelGGL((maxpool_kernel<F32E, IT_F32>), dim3((unsigned)blocks), dim3(NT), 0, stream,
                     reinterpret_cast<const u32x4*>(x), reinterpret_cast<u32x4*>(y), reinterpret_cast<uint2*>(ybf), g);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

int osvos_maxpool2x2_bwd_f32(const float* x, const float* dy, const float* dside, float* dx, void* dxbf,
                             int N, int H, int W, int C, hipStream_t stream) {
  OSVOS_ARG_CHECK(x && dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "maxpool_bwd: bad arguments (C=%d)", C);
  PoolGeo g;
  long blocks;
  OSVOS_ARG_CHECK(make_geo(g, N, H, W, C / 4, NT * IT_F32, blocks) == 0, "maxpool_bwd: tensor too large (%dx%dx%dx%d)", N, H, W, C);
  hipLaunchKernelGGL((maxpool_bwd_kernel<F32E, IT_F32>), dim3((unsigned)blocks), dim3(NT), 0, stream,
                     reinterpret_cast<const u32x4*>(x), reinterpret_cast<const u32x4*>(dy), reinterpret_cast<const u32x4*>(dside),
                     reinterpret_cast<u32x4*>(dx), reinterpret_cast<uint2*>(dxbf), g);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

int osvos_maxpool2x2_bf16(const void* x, void* y, int N, int H, int W, int C, hipStream_t stream) {
  OSVOS_ARG_CHECK(x && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "maxpool bf16: bad arguments (C=%d)", C);
  PoolGeo g;
  long blocks;
  OSVOS_ARG_CHECK(make_geo(g, N, H, W, C / 8, NT * IT_BF16, blocks) == 0, "maxpool bf16: tensor too large (%dx%dx%dx%d)", N, H, W, C);
  hipLaunchKernelGGL((maxpool_kernel<BF16E, IT_BF16>), dim3((unsigned)blocks), dim3(NT), 0, stream,
                     reinterpret_cast<const u32x4*>(x), reinterpret_cast<u32x4*>(y), (uint2*)nullptr, g);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

int osvos_maxpool2x2_bwd_bf16(const void* x, const void* dy, const void* dside, void* dx, int N, int H, int W, int C, hipStream_t stream) {
  OSVOS_ARG_CHECK(x && dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "maxpool_bwd bf16: bad arguments (C=%d)", C);
  PoolGeo g;
  long blocks;
  OSVOS_ARG_CHECK(make_geo(g, N, H, W, C / 8, NT * IT_BF16, blocks) == 0, "maxpool_bwd bf16: tensor too large (%dx%dx%dx%d)", N, H, W, C);
  hipLaunchKernelGGL((maxpool_bwd_kernel<BF16E, IT_BF16>), dim3((unsigned)blocks), dim3(NT), 0, stream,
                     reinterpret_cast<const u32x4*>(x), reinterpret_cast<const u32x4*>(dy), reinterpret_cast<const u32x4*>(dside),
                     reinterpret_cast<u32x4*>(dx), (uint2*)nullptr, g);
  OSVOS_LAUNCH_CHECK();
  return 0;
}
