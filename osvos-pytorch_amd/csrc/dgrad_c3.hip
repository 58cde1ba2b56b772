// Input gradient of conv1_1 (64 -> 3 channels at full resolution): the last data gradient of the backward (reference train_online.py:121
// makes the input a leaf that requires grad, so autograd computes it; vgg_osvos.py:41 first trunk convolution).
//
//   dx[n, ci, y, x] = sum_{r', s', co} dy[n, y + r' - 1, x + s' - 1, co] * W[co, ci, 2 - r', 2 - s']        (zero outside the image)
//
// 1.4 GFLOP over a 105 MB tensor: a bandwidth problem.  As a 32-cout MFMA tile of the f32x3 convolution it wasted 10x the matrix work
// and, one workgroup per CU, took 120-176 us at the tail of the step next to conv1_2's weight gradient (profiles/r03_*timeline*).  Here:
// plain fp32 FMAs, one thread per pixel and its 3 input channels; dy's halo tile goes through LDS in 16-channel chunks as planes of
// 16-byte channel quads ([quad][pixel]: a wave's 64 consecutive pixels read 64 consecutive slots), two buffers, the next chunk's loads in
// flight during the FMAs; the filter never touches LDS or a vector register -- every (tap, channel quad) is 12 consecutive floats of the
// data-gradient pack [tap][co / 4][32][4] (osvos_pack_dgrad_f32), uniform across the wave, i.e. scalar loads feeding v_fma's SGPR operand.
// Result written straight into the caller's NCHW tensor (three planes, lanes = consecutive x).
#include "common.h"
#include "kernels.h"

namespace {

constexpr int TW = 32, TH = 8, HWD = TW + 2, HHT = TH + 2, PLANE = HHT * HWD;      // 256 pixels, 340 halo pixels
constexpr int CQ = 4;                                                             // channel quads per chunk (16 channels)
constexpr int ITEMS = CQ * PLANE, NT = 256, NLD = (ITEMS + NT - 1) / NT;          // 16-byte items per chunk; loads per thread

struct D3Args {
  const float* dy;       // NHWC [N][H][W][Cout], Cout = 64
  const float* wpk;      // data-gradient pack [9][Cout / 4][32][4]
  float* dx;             // NCHW [N][3][H][W]
  int N, H, W, Cout, tiles_x, tiles_y;
};

__global__ __launch_bounds__(NT) void dgrad_c3_kernel(D3Args a) {
  __shared__ f32x4 tile[2][ITEMS];
  const int tid = threadIdx.x;
  int t = blockIdx.x;
  const int tx = t % a.tiles_x;
  t /= a.tiles_x;
  const int ty = t % a.tiles_y, n = t / a.tiles_y;
  const int x0 = tx * TW, y0 = ty * TH;
  const int lx = tid % TW, ly = tid / TW;
  const int nchunks = a.Cout / 16;
  const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy) + (size_t)n * a.H * a.W * a.Cout, 0,
                                                                       (int)((size_t)a.H * a.W * a.Cout * 4), 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  unsigned off[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int e = tid + i * NT;
    const int q = e / PLANE, pix = e % PLANE;
    const int gy = y0 + pix / HWD - 1, gx = x0 + pix % HWD - 1;
    off[i] = (e < ITEMS && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? (unsigned)(((gy * a.W + gx) * a.Cout + 4 * q) * 4) : OOB;
  }
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 reg[NLD];
  auto load = [&](int kc) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) reg[i] = __builtin_amdgcn_raw_buffer_load_b128(drs, off[i], kc * 64, 0);
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NLD; ++i)
      if (ITEMS % NT == 0 || tid + i * NT < ITEMS) tile[buf][tid + i * NT] = __builtin_bit_cast(f32x4, reg[i]);
  };
  float acc[3] = {0.f, 0.f, 0.f};
  load(0);
  store(0);
  __syncthreads();
  for (int kc = 0; kc < nchunks; ++kc) {
    const int buf = kc & 1;
    if (kc + 1 < nchunks) load(kc + 1);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int r = tap / 3, s = tap % 3;
#pragma unroll
      for (int q = 0; q < CQ; ++q) {
        const f32x4 v = tile[buf][q * PLANE + (ly + r) * HWD + lx + s];
        const float* w = a.wpk + ((size_t)(tap * (a.Cout / 4) + kc * CQ + q) * 32) * 4;      // uniform: [ci 0..2][co % 4]
#pragma unroll
        for (int ci = 0; ci < 3; ++ci)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[ci] = __builtin_fmaf(v[e], w[ci * 4 + e], acc[ci]);
      }
    }
    if (kc + 1 < nchunks) {
      store(buf ^ 1);        // (the other buffer was last read in iteration kc - 1, behind the barrier below)
      __syncthreads();
    }
  }
  const int oy = y0 + ly, ox = x0 + lx;
  if (oy < a.H && ox < a.W) {
    float* o = a.dx + ((size_t)n * 3 * a.H + oy) * a.W + ox;
#pragma unroll
    for (int ci = 0; ci < 3; ++ci) o[(size_t)ci * a.H * a.W] = acc[ci];
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
