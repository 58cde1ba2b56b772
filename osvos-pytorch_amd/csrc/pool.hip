// 2x2 stride-2 max-pool with ceil_mode=True, NHWC (reference vgg_osvos.py:140,
// nn.MaxPool2d(kernel_size=2, stride=2, ceil_mode=True) -> aten::max_pool2d_with_indices) and its
// backward fused with the ReLU backward of the producing conv and the side-branch gradient add.
// HBM-bound glue.  fp32 modes: no indices stored -- the backward recomputes the argmax from the saved (post-ReLU) pool input.
// bf16-store mode (round 5 prep): the forward also writes ONE CODE BYTE per pooled element -- bits 1:0 the window position of the first
// maximum in scan order (0,0) (0,1) (1,0) (1,1), bits 5:2 "the input at position q is > 0" -- and the backward reads that byte instead of
// the four inputs: 1 byte instead of 8 per pooled element, nothing else changes (maxpool_bwd_code_kernel; same results bit for bit).
//
// Work item = one 2x2 window x one 16-byte channel group (4 fp32 / 8 bf16 channels).  A workgroup owns a run of consecutive items of
// ONE output row (block -> (image, output row, segment): three block-uniform divisions, everything per lane is 32-bit shifts and adds),
// so a wave's loads are the contiguous even / odd pixels of two input rows and every load of a thread's items is in flight before the
// first one is used.  (The first form flattened (n, oy, ox, c) into one 64-bit index and divided it three times per lane: ~600 VALU
// instructions per 208 bytes moved, a third of the kernel's time at 2.5 TB/s -- profiles/r03_ab_glue_kernels.txt.)
#include "common.h"
#include "kernels.h"

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
// code (optional, bf16 only: E::K == 8): the pool-code byte of every result element (see the header), [N][Ho][Wo][C] bytes
template <class E, int IT>
__global__ __launch_bounds__(NT) void maxpool_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ y, uint2* __restrict__ ybf, uint2* __restrict__ code,
                                                     PoolGeo g) {
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
    unsigned cw[2] = {0u, 0u};
#pragma unroll
    for (int k = 0; k < E::K; ++k) {
      float best = E::get(v[t][0], k);
      unsigned byte = E::get(v[t][0], k) > 0.f ? 4u : 0u;      // (invalid positions hold 0: never > 0, never a strict maximum)
#pragma unroll
      for (int q = 1; q < 4; ++q) {
        const float e = E::get(v[t][q], k);
        if (valid[q] && e > best) { best = e; byte = (byte & ~3u) | (unsigned)q; }
        if (valid[q] && e > 0.f) byte |= 4u << q;
      }
      E::put(m, k, best);      // exact: the winner is an element of the input's own type
      cw[(k >> 2) & 1] |= byte << (8 * (k & 3));
    }
    yr[j[t]] = m;
    if (E::K == 4 && ybf != nullptr) ybf[((size_t)n * g.Ho + oy) * g.row_items + j[t]] = to_bf16x4(m);
    if (E::K == 8 && code != nullptr) code[((size_t)n * g.Ho + oy) * g.row_items + j[t]] = make_uint2(cw[0], cw[1]);
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

// the same from the forward's pool-code bytes (bf16-store mode): dx[pos] = positive(pos) * ( (pos == argmax) * dy + dside[pos] ); the pool input
// itself is not read.  code: [N][Ho][Wo][C] bytes = one uint2 per work item (8 channels).
template <int IT>
__global__ __launch_bounds__(NT) void maxpool_bwd_code_kernel(const uint2* __restrict__ code, const u32x4* __restrict__ dy, const u32x4* __restrict__ dside,
                                                              u32x4* __restrict__ dx, PoolGeo g) {
  typedef BF16E E;
  unsigned n, oy, j0;
  block_coords(g, n, oy, j0, NT * IT);
  const unsigned iy = 2 * oy;
  const bool vy = iy + 1 < (unsigned)g.H;
  const size_t row0 = ((size_t)n * g.H + iy) * g.W * g.CG, row1 = row0 + (size_t)g.W * g.CG;
  const size_t orow = ((size_t)n * g.Ho + oy) * g.row_items;
  const bool side = dside != nullptr;
  u32x4 s[IT][4], gq[IT];
  uint2 cd[IT];
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
    for (int q = 0; q < 4; ++q) s[t][q] = u32x4{0, 0, 0, 0};
    if (live[t]) {
      gq[t] = dy[orow + j];
      cd[t] = code[orow + j];
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
    for (int k = 0; k < 8; ++k) {
      const unsigned byte = (((k >> 2) & 1) ? cd[t].y : cd[t].x) >> (8 * (k & 3));
      const int bi = (int)(byte & 3u);
      const float gk = E::get(gq[t], k);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float r = (q == bi ? gk : 0.f) + E::get(s[t][q], k);
        E::put(out[q], k, ((byte >> (2 + q)) & 1u) ? r : 0.f);
      }
    }
    const size_t at[4] = {row0 + o[t], row0 + o[t] + g.CG, row1 + o[t], row1 + o[t] + g.CG};
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (valid[q]) dx[at[q]] = out[q];
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
                     reinterpret_cast<const u32x4*>(x), reinterpret_cast<u32x4*>(y), reinterpret_cast<uint2*>(ybf), (uint2*)nullptr, g);
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
  return osvos_maxpool2x2_bf16_code(x, y, nullptr, N, H, W, C, stream);
}

// code (optional): [N][Ho][Wo][C] pool-code bytes for osvos_maxpool2x2_bwd_bf16_code
int osvos_maxpool2x2_bf16_code(const void* x, void* y, void* code, int N, int H, int W, int C, hipStream_t stream) {
  OSVOS_ARG_CHECK(x && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "maxpool bf16: bad arguments (C=%d)", C);
  PoolGeo g;
  long blocks;
  OSVOS_ARG_CHECK(make_geo(g, N, H, W, C / 8, NT * IT_BF16, blocks) == 0, "maxpool bf16: tensor too large (%dx%dx%dx%d)", N, H, W, C);
  hipLaunchKernelGGL((maxpool_kernel<BF16E, IT_BF16>), dim3((unsigned)blocks), dim3(NT), 0, stream,
                     reinterpret_cast<const u32x4*>(x), reinterpret_cast<u32x4*>(y), (uint2*)nullptr, reinterpret_cast<uint2*>(code), g);
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

// the backward from the pool-code bytes (written by osvos_maxpool2x2_bf16_code or by the bf16 convolution's fused pool epilogue): x is not needed
int osvos_maxpool2x2_bwd_bf16_code(const void* code, const void* dy, const void* dside, void* dx, int N, int H, int W, int C, hipStream_t stream) {
  OSVOS_ARG_CHECK(code && dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "maxpool_bwd bf16 (code): bad arguments (C=%d)", C);
  PoolGeo g;
  long blocks;
  OSVOS_ARG_CHECK(make_geo(g, N, H, W, C / 8, NT * IT_BF16, blocks) == 0, "maxpool_bwd bf16 (code): tensor too large (%dx%dx%dx%d)", N, H, W, C);
  hipLaunchKernelGGL((maxpool_bwd_code_kernel<IT_BF16>), dim3((unsigned)blocks), dim3(NT), 0, stream, reinterpret_cast<const uint2*>(code),
                     reinterpret_cast<const u32x4*>(dy), reinterpret_cast<const u32x4*>(dside), reinterpret_cast<u32x4*>(dx), g);
  OSVOS_LAUNCH_CHECK();
  return 0;
}
