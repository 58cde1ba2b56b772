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
// BF16IN = 1 (round 4; the bf16-store mode of the network): dy arrives as bf16 NHWC -- half the bytes, one 16-byte load = 8 channels --
// and is widened to fp32 on its way into LDS; filter and arithmetic stay fp32.  Replaces a 32-cout bf16 MFMA tile + layout kernel that
// held the chip for 0.52 ms at batch 12 at the very end of the step (profiles/r03_step_timeline_bf16_b12.txt) for 0.63 GB of reads.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int TW = 32, TH = 16, HWD = TW + 2, HHT = TH + 2, PLANE = HHT * HWD;    // 512 pixels per workgroup, 612 halo pixels
constexpr int CQ = 4;                                                             // channel quads per chunk (16 channels)
constexpr int ITEMS = CQ * PLANE, NT = 256, NLD = (ITEMS + NT - 1) / NT;          // 16-byte items per chunk; loads per thread
constexpr int ITEMS_B = 2 * PLANE, NLD_B = (ITEMS_B + NT - 1) / NT;               // bf16 input: an item = 8 channels = two quads
constexpr int WITEMS = 9 * CQ * 3;                                                // 16-byte filter items per chunk: [tap][quad][ci] x 4 couts

struct D3Args {
  const void* dy;        // NHWC [N][H][W][Cout] fp32 (or bf16: BF16IN), Cout = 64
  const float* wpk;      // data-gradient pack [9][Cout / 4][32][4]
  float* dx;             // NCHW [N][3][H][W]
  int N, H, W, Cout, tiles_x, tiles_y;
};

template <int BF16IN>
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
  constexpr int ES = BF16IN ? 2 : 4, NL = BF16IN ? NLD_B : NLD, NI = BF16IN ? ITEMS_B : ITEMS, CPI = BF16IN ? 8 : 4;      // element size, loads per thread, items, channels per item
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
      if constexpr (BF16IN != 0) {
        if (NI % NT == 0 || e < NI) {      // item (g, pixel): channels 8 g .. 8 g + 7 -> quads 2 g, 2 g + 1 (a bf16 is the upper half of its fp32)
          const int g = e / PLANE, pix = e % PLANE;
          f32x4 lo, hi;
          lo[0] = __uint_as_float(reg[i][0] << 16); lo[1] = __uint_as_float(reg[i][0] & 0xffff0000u);
          lo[2] = __uint_as_float(reg[i][1] << 16); lo[3] = __uint_as_float(reg[i][1] & 0xffff0000u);
          hi[0] = __uint_as_float(reg[i][2] << 16); hi[1] = __uint_as_float(reg[i][2] & 0xffff0000u);
          hi[2] = __uint_as_float(reg[i][3] << 16); hi[3] = __uint_as_float(reg[i][3] & 0xffff0000u);
          tile[(2 * g) * PLANE + pix] = lo;
          tile[(2 * g + 1) * PLANE + pix] = hi;
        }
      } else {
        if (NI % NT == 0 || e < NI) tile[e] = __builtin_bit_cast(f32x4, reg[i]);
      }
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
  hipLaunchKernelGGL(dgrad_c3_kernel<0>, dim3((unsigned)blocks), dim3(NT), 0, stream, a);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

// the same from a bf16 dy (NHWC [N][H][W][Cout] bf16: the bf16-store mode of the network); the filter pack stays the fp32 one
int osvos_conv3x3_dgrad_c3_bf16in(const void* dy_bf16, const float* wpk_dgrad, float* dx_nchw, int N, int H, int W, int Cout, hipStream_t stream) {
  OSVOS_ARG_CHECK(dy_bf16 && wpk_dgrad && dx_nchw && N > 0 && H > 0 && W > 0, "dgrad c3 (bf16 in): bad arguments");
  OSVOS_ARG_CHECK(osvos_dgrad_c3_applicable(3, Cout) && (long)H * W * Cout < (1L << 29), "dgrad c3: Cout %d (multiple of 16) / image too large", Cout);
  D3Args a;
  a.dy = dy_bf16; a.wpk = wpk_dgrad; a.dx = dx_nchw;
  a.N = N; a.H = H; a.W = W; a.Cout = Cout;
  a.tiles_x = ceil_div(W, TW); a.tiles_y = ceil_div(H, TH);
  const long blocks = (long)N * a.tiles_x * a.tiles_y;
  OSVOS_ARG_CHECK(blocks < (1L << 31), "dgrad c3: grid of %ld blocks", blocks);
  hipLaunchKernelGGL(dgrad_c3_kernel<1>, dim3((unsigned)blocks), dim3(NT), 0, stream, a);
  OSVOS_LAUNCH_CHECK();
  return 0;
}
