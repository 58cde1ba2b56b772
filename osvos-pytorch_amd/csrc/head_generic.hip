// GENERIC transposed-convolution head: the path taken when upscale[i].weight is NOT diagonal with one shared filter (a user
// un-froze / re-initialised the 16->16 deconvs; the reference runs arbitrary [16,16,k,k] weights, vgg_osvos.py:46,68).  The fast
// head (head.hip) commutes  fuse o cat o crop o upscale  into "dot-16, then one shared bilinear filter"; that needs the diagonal
// structure interp_surgery writes.  Linearity alone still removes the 64-channel full-resolution concat:
//     fused[Y,X] = b + sum_i sum_taps sum_c P_i[c][y,x] * Weff_i[c][ky][kx],     Weff_i[c][t] = sum_co wfuse[16 i + co] * W_i[c][co][t]
// (the diagonal case is Weff_i[c][t] = wfuse[16 i + c] f[t]).  Backward:
//     dP_i[c][y,x]   = sum_t Weff_i[c][t] dfused[Y(y,t), X(x,t)]  (+ score_dsn part, unchanged)
//     G_i[c][t]      = sum_{n,y,x} P_i[c][n,y,x] dfused[n, Y(y,t), X(x,t)]
//     dwfuse[16i+co] = sum_{c,t} W_i[c][co][t] G_i[c][t]         dW_i[c][co][t] = wfuse[16i+co] G_i[c][t]
// and for the 1->1 side deconvs (arbitrary filter in both paths)  dupscale_[i][t] = sum_pix score_i[pix] dside_i[Y, X].
// Correctness path, not a speed path: plain gathers, double atomics for the tap-indexed reductions.
#include "kernels.h"

namespace {

// weff[c][t] = sum_co wf16[co] * wup[(c * 16 + co) * kk + t]
__global__ void head_weff_kernel(const float* __restrict__ wup, const float* __restrict__ wf16, float* __restrict__ weff, int kk) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 16 * kk) return;
  const int c = idx / kk, t = idx % kk;
  float s = 0.f;
  for (int co = 0; co < 16; ++co) s += wf16[co] * wup[((size_t)c * 16 + co) * kk + t];
  weff[idx] = s;
}

struct UpGArgs {
  const float* score[4];
  const f32x4* prep[4];
  const float* f1[4];
  const float* weff[4];
  const float* fuse_bias;
  float* outs[5];
  int N, H, W;
  int hs[4], ws[4];
};

__global__ void head_upsample_generic_kernel(UpGArgs a) {
  const long hw = (long)a.H * a.W;
  const long total = (long)a.N * hw;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int X = (int)(idx % a.W);
    const int Y = (int)((idx / a.W) % a.H);
    const long n = idx / hw;
    float fused = a.fuse_bias[0];
    for (int i = 0; i < 4; ++i) {
      const int s = 2 << i, k = 2 * s, kk = k * k;
      const int h = a.hs[i], w = a.ws[i];
      const int top = ((h + 1) * s - a.H) / 2, left = ((w + 1) * s - a.W) / 2;
      const int Yp = Y + top, Xp = X + left;
      const int yh = Yp / s, xh = Xp / s;
      float side = 0.f, fu = 0.f;
      for (int ddy = 0; ddy < 2; ++ddy) {
        const int y = yh - 1 + ddy;
        if (y < 0 || y >= h) continue;
        const int ky = Yp - y * s;
        for (int ddx = 0; ddx < 2; ++ddx) {
          const int x = xh - 1 + ddx;
          if (x < 0 || x >= w) continue;
          const int t = ky * k + (Xp - x * s);
          const long li = (n * h + y) * w + x;
          side += a.score[i][li] * a.f1[i][t];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 p = a.prep[i][li * 4 + q];
#pragma unroll
            for (int e = 0; e < 4; ++e) fu += p[e] * a.weff[i][(q * 4 + e) * kk + t];
          }
        }
      }
      a.outs[i][idx] = side;
      fused += fu;
    }
    a.outs[4][idx] = fused;
  }
}

struct HbGArgs {
  const f32x4* prep;
  const float* dside;
  const float* dfused;
  const float* f1;
  const float* weff;
  const float* wd;
  f32x4* dprep;
  uint2* dprep_b;   // optional bf16 copy of dprep (bf16-store mode: what the side_prep data / weight gradients read), [pix][16] bf16
  double* acc;   // per-workgroup partials [gridDim.x][34] in head.hip's layout: [0..15] (zero here: dwfuse comes from G), [16..31] dwd, [32] dbd
  int N, H, W, h, w, s;
};

template <int TPP>
__global__ __launch_bounds__(256) void head_bwd_generic_kernel(HbGArgs a) {
  const int k = 2 * a.s, kk = k * k;
  const int top = ((a.h + 1) * a.s - a.H) / 2, left = ((a.w + 1) * a.s - a.W) / 2;
  const int sub = threadIdx.x % TPP;
  const long npix = (long)a.N * a.h * a.w;
  const int groups_per_block = 256 / TPP;
  float pwd[16], pbd = 0.f;
#pragma unroll
  for (int c = 0; c < 16; ++c) pwd[c] = 0.f;
  const long ngroups_total = (long)gridDim.x * groups_per_block;
  const long iters = (npix + ngroups_total - 1) / ngroups_total;
  for (long it = 0; it < iters; ++it) {
    const long pix = it * ngroups_total + (long)blockIdx.x * groups_per_block + threadIdx.x / TPP;
    const bool live = pix < npix;
    float dfc[16], ds = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) dfc[c] = 0.f;
    if (live) {
      const int x = (int)(pix % a.w), y = (int)((pix / a.w) % a.h);
      const long n = pix / ((long)a.w * a.h);
      for (int t = sub; t < kk; t += TPP) {
        const int ky = t / k, kx = t % k;
        const int Y = y * a.s + ky - top, X = x * a.s + kx - left;
        if (Y >= 0 && Y < a.H && X >= 0 && X < a.W) {
          const long o = (n * a.H + Y) * a.W + X;
          if (a.dfused != nullptr) {
            const float d = a.dfused[o];
#pragma unroll
            for (int c = 0; c < 16; ++c) dfc[c] += a.weff[c * kk + t] * d;
          }
          if (a.dside != nullptr) ds += a.f1[t] * a.dside[o];
        }
      }
    }
#pragma unroll
    for (int o = TPP / 2; o > 0; o >>= 1) {
#pragma unroll
      for (int c = 0; c < 16; ++c) dfc[c] += __shfl_xor(dfc[c], o, 64);
      ds += __shfl_xor(ds, o, 64);
    }
    if (live && sub == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 p = a.prep[pix * 4 + q];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = q * 4 + e;
          o[e] = dfc[c] + a.wd[c] * ds;
          pwd[c] += p[e] * ds;
        }
        a.dprep[pix * 4 + q] = o;
        if (a.dprep_b != nullptr) {
          uint2 hb;
          hb.x = (unsigned)f32_to_bf16(o[0]) | ((unsigned)f32_to_bf16(o[1]) << 16);
          hb.y = (unsigned)f32_to_bf16(o[2]) | ((unsigned)f32_to_bf16(o[3]) << 16);
          a.dprep_b[pix * 4 + q] = hb;
        }
      }
      pbd += ds;
    }
  }
  __shared__ double red[4][34];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const double s2 = wave_sum((double)pwd[c]);
    if (lane == 0) { red[wv][c] = 0.0; red[wv][16 + c] = s2; }
  }
  const double s3 = wave_sum((double)pbd);
  if (lane == 0) { red[wv][32] = s3; red[wv][33] = 0.0; }
  __syncthreads();
  if (threadIdx.x < 34)
    a.acc[(size_t)blockIdx.x * 34 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// G[c][t] += sum over this block's pixels of P[c][pix] * d[n, y s + ky - top, x s + kx - left];  CH = 16 (P = prep) or 1 (P = score)
template <int CH>
__global__ __launch_bounds__(256) void head_tapsum_kernel(const float* __restrict__ P, const float* __restrict__ d, double* __restrict__ G,
                                                          int N, int H, int W, int h, int w, int s) {
  const int k = 2 * s, kk = k * k;
  const int t = blockIdx.x, ky = t / k, kx = t % k;
  const int top = ((h + 1) * s - H) / 2, left = ((w + 1) * s - W) / 2;
  const long npix = (long)N * h * w;
  float g[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) g[c] = 0.f;
  for (long pix = (long)blockIdx.y * blockDim.x + threadIdx.x; pix < npix; pix += (long)gridDim.y * blockDim.x) {
    const int x = (int)(pix % w), y = (int)((pix / w) % h);
    const long n = pix / ((long)w * h);
    const int Y = y * s + ky - top, X = x * s + kx - left;
    if (Y < 0 || Y >= H || X < 0 || X >= W) continue;
    const float v = d[(n * H + Y) * W + X];
#pragma unroll
    for (int c = 0; c < CH; ++c) g[c] += P[pix * CH + c] * v;
  }
  __shared__ double red[4][CH];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const double sum = wave_sum((double)g[c]);
    if (lane == 0) red[wv][c] = sum;
  }
  __syncthreads();
  if (threadIdx.x < CH) atomicAdd(&G[(size_t)threadIdx.x * kk + t], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// dwfuse[co] += sum_{c,t} wup[(c*16+co)*kk + t] * G[c*kk + t]      (one workgroup per co; the fast finalize already wrote / added 0 there)
__global__ __launch_bounds__(256) void head_dwfuse_generic_kernel(const float* __restrict__ wup, const double* __restrict__ G, float* __restrict__ dwf16, int kk) {
  const int co = blockIdx.x;
  double s = 0.0;
  for (int i = threadIdx.x; i < 16 * kk; i += 256) {
    const int c = i / kk, t = i % kk;
    s += (double)wup[((size_t)c * 16 + co) * kk + t] * G[i];
  }
  __shared__ double red[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) dwf16[co] += (float)(red[0] + red[1] + red[2] + red[3]);
}

// dW[c][co][t] (+)= wf16[co] * G[c][t]
__global__ void head_dwup_generic_kernel(const float* __restrict__ wf16, const double* __restrict__ G, float* __restrict__ dw, int kk, int accumulate) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 256 * kk) return;
  const int t = idx % kk, co = (idx / kk) % 16, c = idx / (16 * kk);
  const float v = (float)((double)wf16[co] * G[(size_t)c * kk + t]);
  dw[idx] = accumulate ? dw[idx] + v : v;
}

// dupscale_[t] (+)= G1[t]
__global__ void head_dw1_generic_kernel(const double* __restrict__ G1, float* __restrict__ dw, int kk, int accumulate) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= kk) return;
  dw[idx] = accumulate ? dw[idx] + (float)G1[idx] : (float)G1[idx];
}

inline int grid_for(long total, int cap) {
  long b = (total + 255) / 256;
  return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

}  // namespace

int osvos_head_weff(const float* wup, const float* wf16, float* weff, int k, hipStream_t stream) {
  OSVOS_ARG_CHECK(wup && wf16 && weff && k > 0, "head_weff: bad arguments");
  hipLaunchKernelGGL(head_weff_kernel, dim3((16 * k * k + 255) / 256), dim3(256), 0, stream, wup, wf16, weff, k * k);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

int osvos_head_upsample_generic(const float* const* score, const float* const* prep, const float* const* f1, const float* const* weff,
                                const float* fuse_bias, float* const* outs, int N, int H, int W, const int* hs, const int* ws, hipStream_t stream) {
  OSVOS_ARG_CHECK(score && prep && f1 && weff && fuse_bias && outs && hs && ws && N > 0 && H > 0 && W > 0, "head_upsample_generic: bad arguments");
  UpGArgs a;
  for (int i = 0; i < 4; ++i) {
    a.score[i] = score[i]; a.prep[i] = reinterpret_cast<const f32x4*>(prep[i]); a.f1[i] = f1[i]; a.weff[i] = weff[i];
    a.hs[i] = hs[i]; a.ws[i] = ws[i];
    const int s = 2 << i;
    OSVOS_ARG_CHECK((hs[i] + 1) * s >= H && (ws[i] + 1) * s >= W, "head_upsample_generic: scale %d output smaller than crop", i);
  }
  for (int i = 0; i < 5; ++i) a.outs[i] = outs[i];
  a.fuse_bias = fuse_bias;
  a.N = N; a.H = H; a.W = W;
  hipLaunchKernelGGL(head_upsample_generic_kernel, dim3(grid_for((long)N * H * W, 4096)), dim3(256), 0, stream, a);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

int osvos_head_bwd_generic(const float* prep, const float* dside, const float* dfused, const float* f1, const float* weff, const float* wd,
                           float* dprep, void* dprep_bf16, double* acc, int N, int H, int W, int h, int w, int scale_idx, hipStream_t stream) {
  OSVOS_ARG_CHECK(prep && f1 && weff && wd && dprep && acc, "head_bwd_generic: null pointer");
  OSVOS_ARG_CHECK(scale_idx >= 0 && scale_idx < 4 && N > 0 && H > 0 && W > 0 && h > 0 && w > 0, "head_bwd_generic: bad shape");
  HbGArgs a;
  a.prep = reinterpret_cast<const f32x4*>(prep);
  a.dside = dside; a.dfused = dfused; a.f1 = f1; a.weff = weff; a.wd = wd;
  a.dprep = reinterpret_cast<f32x4*>(dprep);
  a.dprep_b = reinterpret_cast<uint2*>(dprep_bf16);
  a.acc = acc;
  a.N = N; a.H = H; a.W = W; a.h = h; a.w = w; a.s = 2 << scale_idx;
  const int g = osvos_head_bwd_blocks(N, h, w, scale_idx);      // same partial-row count as the fast path (osvos_head_grads_finalize reads it)
  switch (scale_idx) {
    case 0: hipLaunchKernelGGL(head_bwd_generic_kernel<1>, dim3(g), dim3(256), 0, stream, a); break;
    case 1: hipLaunchKernelGGL(head_bwd_generic_kernel<4>, dim3(g), dim3(256), 0, stream, a); break;
    case 2: hipLaunchKernelGGL(head_bwd_generic_kernel<16>, dim3(g), dim3(256), 0, stream, a); break;
    default: hipLaunchKernelGGL(head_bwd_generic_kernel<64>, dim3(g), dim3(256), 0, stream, a); break;
  }
  OSVOS_LAUNCH_CHECK();
  return 0;
}

// G (zeroed here): [channels][k*k] doubles.  channels = 16: P = prep [pix][16]; channels = 1: P = score [pix]
int osvos_head_tapsum(const float* P, int channels, const float* d, double* G, int N, int H, int W, int h, int w, int scale_idx, hipStream_t stream) {
  OSVOS_ARG_CHECK(P && d && G && (channels == 16 || channels == 1) && scale_idx >= 0 && scale_idx < 4, "head_tapsum: bad arguments");
  const int s = 2 << scale_idx, kk = 4 * s * s;
  OSVOS_HIP_CHECK(hipMemsetAsync(G, 0, sizeof(double) * channels * kk, stream));
  const long npix = (long)N * h * w;
  int chunks = (int)((npix + 255) / 256);
  const int cap = kk >= 1024 ? 2 : (kk >= 256 ? 8 : (kk >= 64 ? 32 : 128));      // ~2048 workgroups at most
  if (chunks > cap) chunks = cap;
  if (channels == 16) hipLaunchKernelGGL(head_tapsum_kernel<16>, dim3(kk, chunks), dim3(256), 0, stream, P, d, G, N, H, W, h, w, s);
  else hipLaunchKernelGGL(head_tapsum_kernel<1>, dim3(kk, chunks), dim3(256), 0, stream, P, d, G, N, H, W, h, w, s);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

// dwf16: fuse.weight gradient slice of this scale (+= on top of what osvos_head_grads_finalize left there); dwup: NULL or upscale[i].weight.grad
int osvos_head_generic_param_grads(const float* wup, const float* wf16, const double* G, float* dwf16, float* dwup, int k, int accumulate, hipStream_t stream) {
  OSVOS_ARG_CHECK(wup && wf16 && G && k > 0, "head_generic_param_grads: bad arguments");
  const int kk = k * k;
  if (dwf16 != nullptr) hipLaunchKernelGGL(head_dwfuse_generic_kernel, dim3(16), dim3(256), 0, stream, wup, G, dwf16, kk);
  if (dwup != nullptr) hipLaunchKernelGGL(head_dwup_generic_kernel, dim3(kk), dim3(256), 0, stream, wf16, G, dwup, kk, accumulate);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

int osvos_head_dw1(const double* G1, float* dw, int k, int accumulate, hipStream_t stream) {
  OSVOS_ARG_CHECK(G1 && dw && k > 0, "head_dw1: bad arguments");
  hipLaunchKernelGGL(head_dw1_generic_kernel, dim3((k * k + 255) / 256), dim3(256), 0, stream, G1, dw, k * k, accumulate);
  OSVOS_LAUNCH_CHECK();
  return 0;
}
