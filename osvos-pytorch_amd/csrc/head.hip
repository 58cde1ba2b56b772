// Side-output / fuse head of OSVOS.forward (reference vgg_osvos.py:68-72) in its commuted form.
//
// Reference, per scale i (s = 2^(i+1), k = 2s):   P = side_prep[i](x)            [N,16,h,w]
//   side_out[i] = center_crop(upscale_[i](score_dsn[i](P)))                       1 channel
//   side[i]     = center_crop(upscale[i](P))                                      16 channels
//   fused       = fuse(cat(side))                                                 64 -> 1
// upscale[i].weight is diagonal with ONE shared k x k filter f (interp_surgery,
// osvos_layers.py:72-85; frozen by lr 0 in both training scripts), and upsample, crop and the
// 1x1 fuse are all linear, so     fused = b + sum_i crop(up_f( sum_c wfuse[16i+c] * P[c] )).
// The 64-channel full-resolution concat (105 MB fp32 per 854x480 frame) and the eight negative-pad
// copies (osvos_layers.py:56) never exist here: two dot-16 per low-res pixel, then one gather of
// <= 2x2 taps per scale per output pixel.  The caller verifies the diagonal/shared-filter
// precondition with osvos_deconv_diag_check and refuses to run otherwise.
#include "kernels.h"

namespace {

// score = bd + wd . P[pix], fpart = wf . P[pix]
__global__ void head_lowres_f32_kernel(const f32x4* __restrict__ prep, const float* __restrict__ wd,
                                       const float* __restrict__ bd, const float* __restrict__ wf,
                                       float* __restrict__ score, float* __restrict__ fpart, long npix) {
  float w1[16], w2[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) { w1[c] = wd[c]; w2[c] = wf[c]; }
  const float b = bd[0];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    float s = b, f = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = prep[i * 4 + q];
#pragma unroll
      for (int e = 0; e < 4; ++e) { s += w1[q * 4 + e] * v[e]; f += w2[q * 4 + e] * v[e]; }
    }
    score[i] = s;
    fpart[i] = f;
  }
}

struct UpArgs {
  const float* score[4];
  const float* fpart[4];
  const float* f1[4];
  const float* f16[4];
  const float* fuse_bias;
  float* outs[5];
  int N, H, W;
  int hs[4], ws[4];
};

// transposed conv (k = 2s, stride s, no padding) + center crop, gathered per output pixel:
// out[Y,X] = sum_{y,x} in[y,x] * f[Yp - y*s][Xp - x*s],  Yp = Y + top, top = floor((Ho - H)/2)
// (block = 256 consecutive pixels of ONE image: 32-bit index math; the first form divided a flat 64-bit index three times per pixel)
__global__ __launch_bounds__(256) void head_upsample_kernel(UpArgs a) {
  const unsigned hw = (unsigned)a.H * (unsigned)a.W;
  const unsigned p = blockIdx.x * 256u + threadIdx.x;
  if (p >= hw) return;
  const unsigned n = blockIdx.y;
  const int Y = (int)(p / (unsigned)a.W), X = (int)(p - (unsigned)Y * (unsigned)a.W);
  const size_t idx = (size_t)n * hw + p;
  float fused = a.fuse_bias[0];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int s = 2 << i, k = 2 * s;
    const int h = a.hs[i], w = a.ws[i];
    const int top = ((h + 1) * s - a.H) / 2, left = ((w + 1) * s - a.W) / 2;
    const int Yp = Y + top, Xp = X + left;
    const int yh = Yp / s, xh = Xp / s;
    const float* sc = a.score[i] + (size_t)n * h * w;
    const float* fp = a.fpart[i] + (size_t)n * h * w;
    float side = 0.f, fu = 0.f;
#pragma unroll
    for (int ddy = 0; ddy < 2; ++ddy) {
      const int y = yh - 1 + ddy;
      const int yc = y < 0 ? 0 : (y >= h ? h - 1 : y);
      const int ky = Yp - y * s;
#pragma unroll
      for (int ddx = 0; ddx < 2; ++ddx) {
        const int x = xh - 1 + ddx;
        const int xc = x < 0 ? 0 : (x >= w ? w - 1 : x);
        const int kx = Xp - x * s;
        const bool in = y == yc && x == xc;        // (ky, kx are inside the filter whenever the tap is inside the map)
        const int li = yc * w + xc, t = in ? ky * k + kx : 0;
        // the load is unconditional (clamped address), the SELECT is on the accumulated result: a tap outside the map contributes nothing even
        // when the clamped element is Inf / NaN (0 * Inf would poison the neighbours of a diverged border pixel; the reference never touches those taps)
        side = in ? fmaf(sc[li], a.f1[i][t], side) : side;
        fu = in ? fmaf(fp[li], a.f16[i][t], fu) : fu;
      }
    }
    a.outs[i][idx] = side;
    fused += fu;
  }
  a.outs[4][idx] = fused;
}

struct HbArgs {
  const f32x4* prep;
  const float* dside;
  const float* dfused;
  const float* f1;
  const float* f16;
  const float* wd;
  const float* wf;
  f32x4* dprep;
  uint2* dprep_b;   // optional bf16 copy of dprep (operand of the bf16 side_prep weight gradient)
  double* acc;   // per-workgroup partials [gridDim.x][34]: [0..15] dwf, [16..31] dwd, [32] dbd, [33] spare
  int N, H, W, h, w, s;
};

// One group of TPP lanes per low-resolution pixel: the k*k taps of the transposed-conv adjoint
// (a stride-s gather over the full-resolution upstream gradients) are split across the group and
// reduced with wave shuffles; lane 0 of the group then forms dprep and the parameter-gradient
// partials, which are wave-reduced and pushed with one double atomic per wave per value.
template <int TPP>
__device__ __forceinline__ void head_bwd_body(const HbArgs& a, const unsigned bid, const unsigned nblocks, double (*red)[34]) {
  // the lane-group width fixes the scale: stride S = 2, 4, 8, 16 and K = 2 S taps per axis for TPP = 1, 4, 16, 64 -- every lane owns
  // K * K / TPP = 16 taps (t = sub + j TPP), whose (ky, kx) are shifts of compile-time strides and whose filter values are loaded once per thread
  constexpr int S = TPP == 1 ? 2 : (TPP == 4 ? 4 : (TPP == 16 ? 8 : 16)), K = 2 * S, TAPS = K * K / TPP;
  static_assert(TAPS == 16, "16 taps per lane at every scale");
  const int top = ((a.h + 1) * S - a.H) / 2, left = ((a.w + 1) * S - a.W) / 2;
  const int sub = threadIdx.x % TPP;
  const unsigned hw_lo = (unsigned)a.h * (unsigned)a.w;
  const unsigned npix = (unsigned)a.N * hw_lo;              // (< 2^31: checked on the host)
  constexpr unsigned groups_per_block = 256 / TPP;
  const bool have_f = a.dfused != nullptr, have_s = a.dside != nullptr;
  float w16[TAPS], w1[TAPS];
#pragma unroll
  for (int j = 0; j < TAPS; ++j) {
    w16[j] = a.f16[sub + j * TPP];
    w1[j] = a.f1[sub + j * TPP];
  }
  float pwf[16], pwd[16], pbd = 0.f;
#pragma unroll
  for (int c = 0; c < 16; ++c) { pwf[c] = 0.f; pwd[c] = 0.f; }
  const unsigned ngroups_total = nblocks * groups_per_block;
  const unsigned iters = (npix + ngroups_total - 1) / ngroups_total;
  for (unsigned it = 0; it < iters; ++it) {
    const unsigned pix = it * ngroups_total + bid * groups_per_block + threadIdx.x / TPP;
    const bool live = pix < npix;      // whole groups go dead together; shuffles stay wave-uniform
    float df = 0.f, ds = 0.f;
    unsigned n = 0;
    if (live) {
      n = pix / hw_lo;
      const unsigned r = pix - n * hw_lo;
      const int y = (int)(r / (unsigned)a.w), x = (int)(r - (unsigned)y * (unsigned)a.w);
      const int Y0 = y * S - top, X0 = x * S - left;
      const float* fu = a.dfused + (size_t)n * a.H * a.W;
      const float* sd = a.dside + (size_t)n * a.H * a.W;
      // every tap's load is unconditional (clamped address, the tap dropped by a select outside the frame): all 32 are in flight before the first multiply
      float vf[TAPS], vs[TAPS];
      bool in[TAPS];
#pragma unroll
      for (int j = 0; j < TAPS; ++j) {
        const int t = sub + j * TPP;          // K is a power of two: shifts
        const int Y = Y0 + t / K, X = X0 + t % K;
        const int Yc = Y < 0 ? 0 : (Y >= a.H ? a.H - 1 : Y), Xc = X < 0 ? 0 : (X >= a.W ? a.W - 1 : X);
        in[j] = Y == Yc && X == Xc;
        const int o = Yc * a.W + Xc;
        vf[j] = have_f ? fu[o] : 0.f;
        vs[j] = have_s ? sd[o] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < TAPS; ++j) {
        df = in[j] ? fmaf(w16[j], vf[j], df) : df;      // select on the result, not on the weight: a non-finite clamped element stays out
        ds = in[j] ? fmaf(w1[j], vs[j], ds) : ds;
      }
    }
#pragma unroll
    for (int o = TPP / 2; o > 0; o >>= 1) {
      df += __shfl_xor(df, o, 64);
      ds += __shfl_xor(ds, o, 64);
    }
    if (live && sub == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 p = a.prep[(size_t)pix * 4 + q];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = q * 4 + e;
          o[e] = a.wf[c] * df + a.wd[c] * ds;
          pwf[c] += p[e] * df;
          pwd[c] += p[e] * ds;
        }
        a.dprep[(size_t)pix * 4 + q] = o;
        if (a.dprep_b != nullptr) {
          typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
          bf16x4_t hb;
          hb[0] = (__bf16)o[0]; hb[1] = (__bf16)o[1]; hb[2] = (__bf16)o[2]; hb[3] = (__bf16)o[3];
          a.dprep_b[(size_t)pix * 4 + q] = __builtin_bit_cast(uint2, hb);
        }
      }
      pbd += ds;
    }
  }
  // workgroup partials (no atomics: deterministic, and 2048 waves hammering 33 addresses with
  // double atomics cost ~270 us per scale); osvos_head_grads_finalize sums them
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const double s1 = wave_sum((double)pwf[c]);
    const double s2 = wave_sum((double)pwd[c]);
    if (lane == 0) { red[wv][c] = s1; red[wv][16 + c] = s2; }
  }
  const double s3 = wave_sum((double)pbd);
  if (lane == 0) { red[wv][32] = s3; red[wv][33] = 0.0; }
  __syncthreads();
  if (threadIdx.x < 34)
    a.acc[(size_t)bid * 34 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

template <int TPP>
__global__ __launch_bounds__(256) void head_bwd_f32_kernel(HbArgs a) {
  __shared__ double red[4][34];
  head_bwd_body<TPP>(a, blockIdx.x, gridDim.x, red);
}

// the four scales in ONE launch (they are independent; as four launches of ~20 us each they sit back to back on the backward's critical path):
// workgroups [first[i], first[i + 1]) run scale i's body with its own lane-group width
struct Hb4Args { HbArgs s[4]; unsigned first[5]; };
__global__ __launch_bounds__(256) void head_bwd4_f32_kernel(Hb4Args a) {
  __shared__ double red[4][34];
  const unsigned b = blockIdx.x;
  if (b < a.first[1]) head_bwd_body<1>(a.s[0], b, a.first[1], red);
  else if (b < a.first[2]) head_bwd_body<4>(a.s[1], b - a.first[1], a.first[2] - a.first[1], red);
  else if (b < a.first[3]) head_bwd_body<16>(a.s[2], b - a.first[2], a.first[3] - a.first[2], red);
  else head_bwd_body<64>(a.s[3], b - a.first[3], a.first[4] - a.first[3], red);
}

// per-workgroup partial sums of x -> part[blockIdx.x]
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ x, long count, double* part) {
  __shared__ double red[4];
  double s = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) s += (double)x[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// out[0] = max |w[ci][co]| over ci != co, out[1] = max |w[c][c] - w[0][0]|   (w: [C][C][k][k])
__global__ void deconv_diag_check_kernel(const float* __restrict__ w, int C, int kk, float* out) {
  float off = 0.f, dev = 0.f;
  const long total = (long)C * C * kk;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int t = (int)(i % kk);
    const int co = (int)((i / kk) % C);
    const int ci = (int)(i / ((long)kk * C));
    const float v = w[i];
    if (ci != co) off = fmaxf(off, fabsf(v));
    else dev = fmaxf(dev, fabsf(v - w[t]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    off = fmaxf(off, __shfl_xor(off, o, 64));
    dev = fmaxf(dev, __shfl_xor(dev, o, 64));
  }
  if ((threadIdx.x & 63) == 0) {   // values are >= 0: integer max on the bit pattern is order preserving
    atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint(off));
    atomicMax(reinterpret_cast<unsigned int*>(out) + 1, __float_as_uint(dev));
  }
}

inline int grid_for(long total, int cap) {
  long b = (total + 255) / 256;
  return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

}  // namespace

int osvos_head_lowres_f32(const float* prep, const float* wd, const float* bd, const float* wf,
                          float* score, float* fpart, int N, int h, int w, hipStream_t stream) {
  OSVOS_ARG_CHECK(prep && wd && bd && wf && score && fpart && N > 0 && h > 0 && w > 0, "head_lowres: bad arguments");
  const long npix = (long)N * h * w;
  hipLaunchKernelGGL(head_lowres_f32_kernel, dim3(grid_for(npix, 2048)), dim3(256), 0, stream,
                     reinterpret_cast<const f32x4*>(prep), wd, bd, wf, score, fpart, npix);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

extern "C" int osvos_head_upsample(const float* const* score, const float* const* fpart,
                                   const float* const* f1, const float* const* f16, const float* fuse_bias,
                                   float* const* outs, int N, int H, int W, const int* hs, const int* ws, void* stream) {
  OSVOS_ARG_CHECK(score && fpart && f1 && f16 && fuse_bias && outs && hs && ws && N > 0 && H > 0 && W > 0, "head_upsample: bad arguments");
  UpArgs a;
  for (int i = 0; i < 4; ++i) {
    a.score[i] = score[i]; a.fpart[i] = fpart[i]; a.f1[i] = f1[i]; a.f16[i] = f16[i];
    a.hs[i] = hs[i]; a.ws[i] = ws[i];
    const int s = 2 << i;
    OSVOS_ARG_CHECK((hs[i] + 1) * s >= H && (ws[i] + 1) * s >= W, "head_upsample: scale %d output smaller than crop", i);
  }
  for (int i = 0; i < 5; ++i) a.outs[i] = outs[i];
  a.fuse_bias = fuse_bias;
  a.N = N; a.H = H; a.W = W;
  OSVOS_ARG_CHECK((long)H * W < (1L << 31) && N <= 65535, "head_upsample: frame / batch too large (%d x %d x %d)", N, H, W);
  hipLaunchKernelGGL(head_upsample_kernel, dim3((unsigned)(((long)H * W + 255) / 256), (unsigned)N), dim3(256), 0, (hipStream_t)stream, a);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

int osvos_head_bwd_blocks(int N, int h, int w, int scale_idx) {
  const int tpp[4] = {1, 4, 16, 64};
  const long npix = (long)N * h * w;
  return grid_for(npix * tpp[scale_idx], OSVOS_HEAD_MAX_BLOCKS);
}

int osvos_head_bwd_f32(const float* prep, const float* dside, const float* dfused, const float* f1, const float* f16,
                       const float* wd, const float* wf, float* dprep, void* dprep_bf16, double* acc, int N, int H, int W, int h, int w,
                       int scale_idx, hipStream_t stream) {
  OSVOS_ARG_CHECK(prep && f1 && f16 && wd && wf && dprep && acc, "head_bwd: null pointer");
  OSVOS_ARG_CHECK(scale_idx >= 0 && scale_idx < 4 && N > 0 && H > 0 && W > 0 && h > 0 && w > 0, "head_bwd: bad shape");
  HbArgs a;
  a.prep = reinterpret_cast<const f32x4*>(prep);
  a.dside = dside; a.dfused = dfused; a.f1 = f1; a.f16 = f16; a.wd = wd; a.wf = wf;
  a.dprep = reinterpret_cast<f32x4*>(dprep);
  a.dprep_b = reinterpret_cast<uint2*>(dprep_bf16);
  a.acc = acc;
  a.N = N; a.H = H; a.W = W; a.h = h; a.w = w; a.s = 2 << scale_idx;
  const int g = osvos_head_bwd_blocks(N, h, w, scale_idx);
  switch (scale_idx) {
    case 0: hipLaunchKernelGGL(head_bwd_f32_kernel<1>, dim3(g), dim3(256), 0, stream, a); break;
    case 1: hipLaunchKernelGGL(head_bwd_f32_kernel<4>, dim3(g), dim3(256), 0, stream, a); break;
    case 2: hipLaunchKernelGGL(head_bwd_f32_kernel<16>, dim3(g), dim3(256), 0, stream, a); break;
    default: hipLaunchKernelGGL(head_bwd_f32_kernel<64>, dim3(g), dim3(256), 0, stream, a); break;
  }
  OSVOS_LAUNCH_CHECK();
  return 0;
}

// all four scales in one launch: arrays indexed by scale; same partial layout per scale as osvos_head_bwd_f32 (acc[i]: osvos_head_bwd_blocks x 34)
int osvos_head_bwd4_f32(const float* const* prep, const float* const* dside, const float* dfused, const float* const* f1, const float* const* f16,
                        const float* const* wd, const float* wf, float* const* dprep, void* const* dprep_bf16, double* const* acc,
                        int N, int H, int W, const int* hs, const int* ws, hipStream_t stream) {
  Hb4Args a;
  a.first[0] = 0;
  for (int i = 0; i < 4; ++i) {
    OSVOS_ARG_CHECK(prep[i] && f1[i] && f16[i] && wd[i] && wf && dprep[i] && acc[i] && hs[i] > 0 && ws[i] > 0, "head_bwd4: bad arguments for scale %d", i);
    HbArgs& q = a.s[i];
    q.prep = reinterpret_cast<const f32x4*>(prep[i]);
    q.dside = dside[i]; q.dfused = dfused; q.f1 = f1[i]; q.f16 = f16[i]; q.wd = wd[i]; q.wf = wf + 16 * i;
    q.dprep = reinterpret_cast<f32x4*>(dprep[i]);
    q.dprep_b = reinterpret_cast<uint2*>(dprep_bf16 ? dprep_bf16[i] : nullptr);
    q.acc = acc[i];
    q.N = N; q.H = H; q.W = W; q.h = hs[i]; q.w = ws[i]; q.s = 2 << i;
    a.first[i + 1] = a.first[i] + (unsigned)osvos_head_bwd_blocks(N, hs[i], ws[i], i);
  }
  hipLaunchKernelGGL(head_bwd4_f32_kernel, dim3(a.first[4]), dim3(256), 0, stream, a);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

int osvos_sum_partials(const float* x, long count, double* part, int* nblocks, hipStream_t stream) {
  const int g = grid_for(count, OSVOS_HEAD_MAX_BLOCKS);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(g), dim3(256), 0, stream, x, count, part);
  OSVOS_LAUNCH_CHECK();
  *nblocks = g;
  return 0;
}

extern "C" int osvos_deconv_diag_check(const float* w, int C, int k, float* out2, void* stream) {
  OSVOS_ARG_CHECK(w && out2 && C > 0 && k > 0, "deconv_diag_check: bad arguments");
  OSVOS_HIP_CHECK(hipMemsetAsync(out2, 0, 2 * sizeof(float), (hipStream_t)stream));
  hipLaunchKernelGGL(deconv_diag_check_kernel, dim3(grid_for((long)C * C * k * k, 256)), dim3(256), 0, (hipStream_t)stream, w, C, k * k, out2);
  OSVOS_LAUNCH_CHECK();
  return 0;
}
