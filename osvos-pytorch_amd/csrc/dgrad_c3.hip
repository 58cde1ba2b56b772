// Input gradient of conv1_1 (64 -> 3 channels at full resolution): the last data gradient of the backward (reference train_online.py:121
// makes the input a leaf that requires grad, so autograd computes it; vgg_osvos.py:41 first trunk convolution).
//
//   dx[n, ci, y, x] = sum_{r', s', co} dy[n, y + r' - 1, x + s' - 1, co] * W[co, ci, 2 - r', 2 - s']        (zero outside the image)
//
// 1.4 GFLOP over a 105 MB tensor: a bandwidth problem.  As a 32-cout MFMA tile of the f32x3 convolution it wasted 10x the matrix work
// and, one workgroup per CU, took 120-176 us at the tail of the step next to conv1_2's weight gradient (profiles/r03_*timeline*).  Here:
// plain fp32 FMAs, one thread per TWO pixels (rows y, y + 8 of a 32 x 16 tile) and their 3 input channels; per 16-channel chunk dy's
// halo tile sits in LDS as planes of 16-byte channel quads ([quad][pixel]: a wave's 64 consecutive pixels read 64 consecutive slots) and
// the chunk's filter slice (9 taps x 4 quads x 12 floats of the data-gradient pack [tap][co / 4][32][4], osvos_pack_dgrad_f32) next to it,
// read as wave-wide broadcasts: 2 + 3 LDS reads per 24 FMAs.  40 KB of LDS and <= 128 registers: four workgroups per CU cover each
// other's load / barrier phases.  (A first form fed the filter through scalar loads: every (tap, quad) then waited out an s_load --
// 202 us per launch.)  Result written straight into the caller's NCHW tensor (three planes, lanes = consecutive x).
// (The bf16-store mode of the network has its own kernel below: the whole halo tile once through LDS, MFMA.)
#include "common.h"
#include "kernels.h"

namespace {

constexpr int TW = 32, TH = 16, HWD = TW + 2, HHT = TH + 2, PLANE = HHT * HWD;    // 512 pixels per workgroup, 612 halo pixels
constexpr int CQ = 4;                                                             // channel quads per chunk (16 channels)
constexpr int ITEMS = CQ * PLANE, NT = 256, NLD = (ITEMS + NT - 1) / NT;          // 16-byte items per chunk; loads per thread
constexpr int WITEMS = 9 * CQ * 3;                                                // 16-byte filter items per chunk: [tap][quad][ci] x 4 couts

struct D3Args {
  const float* dy;       // NHWC [N][H][W][Cout] fp32
  const float* wpk;      // data-gradient pack [9][Cout / 4][32][4]
  float* dx;             // NCHW [N][3][H][W]
  int N, H, W, Cout, tiles_x, tiles_y;
};

__global__ __launch_bounds__(NT, 4) void dgrad_c3_kernel(D3Args a) {
  __shared__ f32x4 tile[ITEMS];
  __shared__ f32x4 wl[WITEMS];
  const int tid = threadIdx.x;
  int t = blockIdx.x;
  const int tx = t % a.tiles_x;
  t /= a.tiles_x;
  const int ty = t % a.tiles_y, n = t / a.tiles_y;
  const int x0 = tx * TW, y0 = ty * TH;
  const int lx = tid % TW, ly = tid / TW;              // this thread's pixels: (ly, lx) and (ly + 8, lx)
  const int nchunks = a.Cout / 16;
  constexpr int ES = 4, NL = NLD, NI = ITEMS, CPI = 4;      // element size, loads per thread, items, channels per item
  const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(a.dy)) + (size_t)n * a.H * a.W * a.Cout * ES, 0,
                                                                       (int)((size_t)a.H * a.W * a.Cout * ES), 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  unsigned off[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int e = tid + i * NT;
    const int q = e / PLANE, pix = e % PLANE;
    const int gy = y0 + pix / HWD - 1, gx = x0 + pix % HWD - 1;
    off[i] = (e < NI && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? (unsigned)(((gy * a.W + gx) * a.Cout + CPI * q) * ES) : OOB;
  }
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  // filter item of this thread (tid < WITEMS): (tap, quad, ci) -> 4 consecutive couts of the pack
  const int wtap = tid / (CQ * 3), wq = (tid / 3) % CQ, wci = tid % 3;
  float acc[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
  for (int kc = 0; kc < nchunks; ++kc) {
    u32x4 reg[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) reg[i] = __builtin_amdgcn_raw_buffer_load_b128(drs, off[i], kc * 16 * ES, 0);
    f32x4 wreg = {0.f, 0.f, 0.f, 0.f};
    if (tid < WITEMS) wreg = *reinterpret_cast<const f32x4*>(a.wpk + ((size_t)(wtap * (a.Cout / 4) + kc * CQ + wq) * 32 + wci) * 4);
    __syncthreads();          // everybody is done with the previous chunk's tiles
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int e = tid + i * NT;
      if (NI % NT == 0 || e < NI) tile[e] = __builtin_bit_cast(f32x4, reg[i]);
    }
    if (tid < WITEMS) wl[tid] = wreg;
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int r = tap / 3, s = tap % 3;
#pragma unroll
      for (int q = 0; q < CQ; ++q) {
        const f32x4 v0 = tile[q * PLANE + (ly + r) * HWD + lx + s];
        const f32x4 v1 = tile[q * PLANE + (ly + 8 + r) * HWD + lx + s];
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
          const f32x4 w = wl[(tap * CQ + q) * 3 + ci];          // same address in every lane: a broadcast read
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[0][ci] = __builtin_fmaf(v0[e], w[e], acc[0][ci]);
            acc[1][ci] = __builtin_fmaf(v1[e], w[e], acc[1][ci]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int oy = y0 + ly + 8 * h, ox = x0 + lx;
    if (oy < a.H && ox < a.W) {
      float* o = a.dx + ((size_t)n * 3 * a.H + oy) * a.W + ox;
#pragma unroll
      for (int ci = 0; ci < 3; ++ci) o[(size_t)ci * a.H * a.W] = acc[h][ci];
    }
  }
}


// ---- the same gradient from a bf16 dy on the matrix pipe (round 4; the bf16-store mode of the network, batch 12) ----------------------------
// 630 MB of dy in, 59 MB out: a stream.  The 32-cout bf16 tile of the general convolution + a layout kernel took 0.52 ms at the very end of the
// step (profiles/r03_step_timeline_bf16_b12.txt); the fp32-FMA kernel above fed with bf16 (a first attempt of this round) was FMA bound and
// re-fetched every 128-byte pixel line once per 16-channel chunk (2.9 GB fetched for 0.63 GB).  Here a workgroup loads the WHOLE 64-channel halo tile once -- 34 x 18
// pixels x 128 bytes, every pixel one full line, 78 KB of LDS as eight planes of 8-channel groups -- and its four waves run
// v_mfma_f32_32x32x16_bf16 with the filter as the row operand (rows = the 3 input channels, zero padded to 32: the pack of the general bf16
// data gradient, [tap][channel group][32] x 8 bf16, read by the three lanes that have a row) and 32 consecutive pixels of an image row as
// columns: 36 MFMAs per row, the 36 filter fragments in registers, two rows in flight.  10x the useful matrix work and still ~70 us of MFMA
// time at batch 12 against ~170 us of HBM time.  Lanes 0..31 hold (pixel, ci = 0..2) in accumulator registers 0..2: three 128-byte NCHW stores.
constexpr int M_PIX = HHT * HWD, M_PLANE = M_PIX + 1;          // 612 halo pixels; planes one slot apart in bank phase
constexpr int M_ITEMS = M_PIX * 8, M_NLD = (M_ITEMS + NT - 1) / NT;
constexpr int M_LDS_BYTES = 8 * M_PLANE * 16;

struct D3mArgs {
  const void* dy;        // NHWC [N][H][W][64] bf16
  const uint4* wpk;      // bf16 data-gradient pack [9][8][32] x (8 bf16)
  float* dx;             // NCHW [N][3][H][W]
  int N, H, W, tiles_x, tiles_y;
};

__global__ __launch_bounds__(NT, 2) void dgrad_c3_mfma_kernel(D3mArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_m[];
  uint4* As = reinterpret_cast<uint4*>(smem_m);
  typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  int t = blockIdx.x;
  const int tx = t % a.tiles_x;
  t /= a.tiles_x;
  const int ty = t % a.tiles_y, n = t / a.tiles_y;
  const int x0 = tx * TW, y0 = ty * TH;
  const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(a.dy)) + (size_t)n * a.H * a.W * 128, 0,
                                                                       (int)((size_t)a.H * a.W * 128), 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  u32x4 reg[M_NLD];
#pragma unroll
  for (int i = 0; i < M_NLD; ++i) {      // item e = (halo pixel, 8-channel group): consecutive lanes walk a pixel's 128 bytes, then the next pixel of the row
    const int e = tid + i * NT;
    const int g = e & 7, pix = e >> 3;
    const int gy = y0 + pix / HWD - 1, gx = x0 + pix % HWD - 1;
    const unsigned off = (e < M_ITEMS && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? (unsigned)((gy * a.W + gx) * 128 + g * 16) : OOB;
    reg[i] = __builtin_amdgcn_raw_buffer_load_b128(drs, off, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < M_NLD; ++i) {
    const int e = tid + i * NT;
    if (e < M_ITEMS) As[(e & 7) * M_PLANE + (e >> 3)] = __builtin_bit_cast(uint4, reg[i]);
  }
  // the 36 filter fragments of this lane: row li of the (tap, channel group 2 ks + lh) block; rows 3..31 are zero without being read (a
  // buffer load past num_records returns zeros: no branch, no second copy of the 144 registers).  Issued once the tile has left its 80
  // staging registers -- the two sets together spill -- and in flight across the barrier.
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(a.wpk), 0, 9 * 8 * 32 * 16, 0x00020000);
  const unsigned woff = li < 3 ? (unsigned)((lh * 32 + li) * 16) : OOB;
  u32x4 wf[36];
#pragma unroll
  for (int st = 0; st < 36; ++st) wf[st] = __builtin_amdgcn_raw_buffer_load_b128(wrs, woff, ((st >> 2) * 8 + 2 * (st & 3)) * 32 * 16, 0);
  __syncthreads();
  // wave w: rows 4 w .. 4 w + 3 of the tile, two at a time (two independent accumulator chains)
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int row = wave * 4 + half * 2;
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int st = 0; st < 36; ++st) {
      const int tap = st >> 2, ks = st & 3;
      const int r = tap / 3, s2 = tap % 3;
      const uint4* base = As + (2 * ks + lh) * M_PLANE + (row + r) * HWD + li + s2;
      const uint4 f0 = base[0], f1 = base[HWD];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf[st]), __builtin_bit_cast(bf16x8_t, f0), acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf[st]), __builtin_bit_cast(bf16x8_t, f1), acc[1], 0, 0, 0);
    }
    if (lh == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int oy = y0 + row + j, ox = x0 + li;
        if (oy < a.H && ox < a.W) {
          float* o = a.dx + ((size_t)n * 3 * a.H + oy) * a.W + ox;
#pragma unroll
          for (int ci = 0; ci < 3; ++ci) o[(size_t)ci * a.H * a.W] = acc[j][ci];
        }
      }
    }
  }
}

}  // namespace

bool osvos_dgrad_c3_applicable(int Cin, int Cout) { return Cin == 3 && Cout % 16 == 0 && Cout >= 16; }

// dy: NHWC fp32 [N][H][W][Cout] (already ReLU-masked); wpk_dgrad: osvos_pack_dgrad_f32 pack of the [Cout][3][3][3] filter; dx_nchw: [N][3][H][W]
int osvos_conv3x3_dgrad_c3_f32(const float* dy, const float* wpk_dgrad, float* dx_nchw, int N, int H, int W, int Cout, hipStream_t stream) {
  OSVOS_ARG_CHECK(dy && wpk_dgrad && dx_nchw && N > 0 && H > 0 && W > 0, "dgrad c3: bad arguments");
  OSVOS_ARG_CHECK(osvos_dgrad_c3_applicable(3, Cout) && (long)H * W * Cout < (1L << 29), "dgrad c3: Cout %d (multiple of 16) / image too large", Cout);
  D3Args a;
  a.dy = dy; a.wpk = wpk_dgrad; a.dx = dx_nchw;
  a.N = N; a.H = H; a.W = W; a.Cout = Cout;
  a.tiles_x = ceil_div(W, TW); a.tiles_y = ceil_div(H, TH);
  const long blocks = (long)N * a.tiles_x * a.tiles_y;
  OSVOS_ARG_CHECK(blocks < (1L << 31), "dgrad c3: grid of %ld blocks", blocks);
  hipLaunchKernelGGL(dgrad_c3_kernel, dim3((unsigned)blocks), dim3(NT), 0, stream, a);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

// dy: NHWC bf16 [N][H][W][64]; wpk_bf16_dgrad: the bf16 data-gradient pack of the [64][3][3][3] filter (osvos_pack_dgrad with the bf16 dtype:
// [9][8][32] entries of 8 bf16); dx_nchw: [N][3][H][W] fp32.  bf16 operands, fp32 accumulation (the arithmetic of the general bf16 convolution).
int osvos_conv3x3_dgrad_c3_bf16mfma(const void* dy_bf16, const void* wpk_bf16_dgrad, float* dx_nchw, int N, int H, int W, int Cout, hipStream_t stream) {
  OSVOS_ARG_CHECK(dy_bf16 && wpk_bf16_dgrad && dx_nchw && N > 0 && H > 0 && W > 0, "dgrad c3 (bf16 mfma): bad arguments");
  OSVOS_ARG_CHECK(Cout == 64 && (long)H * W * 128 < (1L << 31), "dgrad c3 (bf16 mfma): Cout %d (64 only) / image too large", Cout);
  static bool attr_set_dev[OSVOS_MAX_DEVICES] = {};
  bool& attr_set = attr_set_dev[osvos_current_device()];
  if (!attr_set) {
    OSVOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dgrad_c3_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, M_LDS_BYTES));
    attr_set = true;
  }
  D3mArgs a;
  a.dy = dy_bf16; a.wpk = reinterpret_cast<const uint4*>(wpk_bf16_dgrad); a.dx = dx_nchw;
  a.N = N; a.H = H; a.W = W;
  a.tiles_x = ceil_div(W, TW); a.tiles_y = ceil_div(H, TH);
  const long blocks = (long)N * a.tiles_x * a.tiles_y;
  OSVOS_ARG_CHECK(blocks < (1L << 31), "dgrad c3 (bf16 mfma): grid of %ld blocks", blocks);
  hipLaunchKernelGGL(dgrad_c3_mfma_kernel, dim3((unsigned)blocks), dim3(NT), M_LDS_BYTES, stream, a);
  OSVOS_LAUNCH_CHECK();
  return 0;
}
