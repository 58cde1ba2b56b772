// Class-balanced binary cross-entropy with logits, forward + gradient in one pass over the map
// (reference layers/osvos_layers.py:19-48; called train_online.py:127, train_parent.py:145).
//
//   y = label >= 0.5;  n_pos = sum y;  w_pos = n_neg/n_tot, w_neg = n_pos/n_tot  (float32 quotients,
//   as in the reference where both masks are .float());
//   val = x*(y - [x>=0]) - log(1 + exp(x - 2x[x>=0]))          (== y*x - softplus(x), stable form)
//   loss = (w_pos * sum(-y*val) + w_neg * sum(-(1-y)*val)) / div,   dLoss/dx = w(y) (sigmoid(x) - y) / div
// HBM-bound: two sweeps of the logit map (count, then loss+grad); wavefront shuffles + one double
// atomic per wave; everything stays on the device (no .item() needed to form the loss).
#include "common.h"

namespace {

// Scratch of one call: npos[G] (unsigned long long), then lpos[H][G], lneg[H][G] (doubles), then the arrival ticket (8 bytes) -- G = count groups (1,
// or the N images in the per-image mode), H = heads.  8 G + 16 H G + 8 bytes (osvos_cbce_scratch_bytes); with G = 1 that fits the 32 bytes per
// head the first form of the interface asked for.  Zero on entry (zeroed by the call, or by the caller's promise: OSVOS_CBCE_SCRATCH_ZEROED) and
// ALWAYS left zero on exit: the workgroup that arrives last forms the losses and clears what it read (round 6: the loss call is two launches --
// count, sweep -- instead of memset (two fill kernels for a 32-byte region) + count + sweep + final: 114 -> ~90 us between forward and backward).
struct ScratchView {
  unsigned long long* npos;
  double* lpos;
  double* lneg;
  unsigned* ticket;
};
__device__ __host__ inline ScratchView scratch_view(void* p, int G, int H) {
  ScratchView v;
  v.npos = reinterpret_cast<unsigned long long*>(p);
  v.lpos = reinterpret_cast<double*>(v.npos + G);
  v.lneg = v.lpos + (size_t)H * G;
  v.ticket = reinterpret_cast<unsigned*>(v.lneg + (size_t)H * G);
  return v;
}

// (both sweeps read 16 bytes per lane and load over the first 4 * n4 elements of a group, element-wise over the rest: n4 = elements / 4 for
//  16-byte aligned tensors -- the rest is then at most 3 elements -- and 0 for tensors that are not: a contiguous but offset view such as
//  outputs[-1][1:2] with H * W % 4 != 0 takes the scalar sweep instead of being refused)
__global__ void cbce_count_kernel(const float* __restrict__ label, long per_group, long n4, unsigned long long* npos) {
  const int grp = blockIdx.y;
  const float* lab = label + (size_t)grp * per_group;
  unsigned int c = 0;
  const f32x4* l4 = reinterpret_cast<const f32x4*>(lab);
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long)gridDim.x * blockDim.x;
  for (long i = tid; i < n4; i += nth) {
    const f32x4 v = l4[i];
    c += (v[0] >= 0.5f ? 1u : 0u) + (v[1] >= 0.5f ? 1u : 0u) + (v[2] >= 0.5f ? 1u : 0u) + (v[3] >= 0.5f ? 1u : 0u);
  }
  for (long i = 4 * n4 + tid; i < per_group; i += nth) c += lab[i] >= 0.5f ? 1u : 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  __shared__ unsigned int red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    c = red[0] + red[1] + red[2] + red[3];
    if (c) atomicAdd(&npos[grp], (unsigned long long)c);     // one atomic per workgroup
  }
}

// one element: accumulates the two loss sums, returns the gradient
__device__ __forceinline__ float cbce_elem(float x, float lab, float wpos, float wneg, float inv_div, float gscale, bool want_grad, double& lpos, double& lneg) {
  const float y = lab >= 0.5f ? 1.f : 0.f;
  const float g = x >= 0.f ? 1.f : 0.f;
  const float val = x * (y - g) - logf(1.f + expf(x - 2.f * x * g));
  lpos += (double)(-y * val);
  lneg += (double)(-(1.f - y) * val);
  if (!want_grad) return 0.f;
  // d(-val)/dx = (1 - 2g) * sigmoid(-|x|) - (y - g): the form autograd derives from the
  // reference's expression; no cancellation for saturated logits (sigmoid(x) - y would lose it)
  const float ez = expf(-fabsf(x));
  const float sz = ez / (1.f + ez);
  const float gi = (y > 0.5f ? wpos : wneg) * ((1.f - 2.f * g) * sz - (y - g)) * inv_div;
  // gscale: the upstream gradient of the loss (1 / nAveGrad, times the side-head weight in the parent loop), applied to the
  // ROUNDED per-pixel gradient -- the same two roundings as this kernel followed by osvos_scale (autograd's chain)
  return gi * gscale;
}

constexpr int kMaxHeads = 8;
struct CbceHeads {
  const float* out[kMaxHeads];
  float* grad[kMaxHeads];
  float* loss[kMaxHeads];
  float* running[kMaxHeads];
  float gscale[kMaxHeads];
  int n;
};

// What a call weights and divides with (osvos_layers.py:28-34,43-46), per count group `grp`:
//   default          counts of the whole input tensor, size_average / count, batch_average / N
//   per image        counts of image `grp` alone, size_average / (elements of one image), batch_average / 1: every image is its own
//                    reference batch of one (the micro-batches of an accumulation window, train_online.py:116-149, in one call)
//   external counts  {n_pos, n_total, n_images} of the GLOBAL batch this call holds a shard of (device floats from the count exchange of a
//                    sharded batch): weights and divisors from them
struct CbceNorm {
  const float* counts;      // external counts or NULL
  int mode;                 // 0 size_average, 1 batch_average, 2 neither
  int N;                    // images of this call
  int per_image;
};
__device__ __forceinline__ void cbce_weights(const CbceNorm& nm, const unsigned long long* npos_grp, int grp, long per_group, float& wpos, float& wneg, float& inv_div) {
  float ntot, npos, nimg;
  if (nm.counts != nullptr) { npos = nm.counts[0]; ntot = nm.counts[1]; nimg = nm.counts[2]; }
  else { npos = (float)npos_grp[grp]; ntot = (float)per_group; nimg = nm.per_image ? 1.f : (float)nm.N; }
  wpos = (ntot - npos) / ntot;
  wneg = npos / ntot;
  inv_div = nm.mode == 0 ? 1.f / ntot : (nm.mode == 1 ? 1.f / nimg : 1.f);
}

// grid (workgroups, count groups, heads)
__global__ void cbce_main_kernel(CbceHeads hd, const float* __restrict__ label, long per_group, long n4, CbceNorm nm, ScratchView sc, int G) {
  const int grp = blockIdx.y, head = blockIdx.z;
  float wpos, wneg, inv_div;
  cbce_weights(nm, sc.npos, grp, per_group, wpos, wneg, inv_div);
  const float* __restrict__ out = hd.out[head] + (size_t)grp * per_group;
  const float* __restrict__ lab = label + (size_t)grp * per_group;
  float* __restrict__ grad = hd.grad[head] != nullptr ? hd.grad[head] + (size_t)grp * per_group : nullptr;
  const float gscale = hd.gscale[head];
  const bool want = grad != nullptr;
  double lpos = 0.0, lneg = 0.0;
  const f32x4* o4 = reinterpret_cast<const f32x4*>(out);
  const f32x4* l4 = reinterpret_cast<const f32x4*>(lab);
  f32x4* g4 = reinterpret_cast<f32x4*>(grad);
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long)gridDim.x * blockDim.x;
  for (long i = tid; i < n4; i += nth) {
    const f32x4 x = o4[i], lb = l4[i];
    f32x4 g;
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = cbce_elem(x[e], lb[e], wpos, wneg, inv_div, gscale, want, lpos, lneg);
    if (want) g4[i] = g;
  }
  for (long i = 4 * n4 + tid; i < per_group; i += nth) {
    const float g = cbce_elem(out[i], lab[i], wpos, wneg, inv_div, gscale, want, lpos, lneg);
    if (want) grad[i] = g;
  }
  lpos = wave_sum(lpos);
  lneg = wave_sum(lneg);
  __shared__ double red[4][2];
  __shared__ int last;
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = lpos; red[threadIdx.x >> 6][1] = lneg; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&sc.lpos[(size_t)head * G + grp], red[0][0] + red[1][0] + red[2][0] + red[3][0]);      // (device-scope RMWs at the L2)
    atomicAdd(&sc.lneg[(size_t)head * G + grp], red[0][1] + red[1][1] + red[2][1] + red[3][1]);
    // arrival ticket: release the two sums, draw; the workgroup that draws the last number finishes the call (cdna_hip_programming.md 6 G16:
    // agent-scope release before the ticket, agent-scope acquire in the last arriver, relaxed everything else)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned total = gridDim.x * gridDim.y * gridDim.z;
    last = __hip_atomic_fetch_add(sc.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == total - 1u ? 1 : 0;
  }
  __syncthreads();
  if (!last) return;
  if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  __syncthreads();
  // one thread per head: loss = sum over the count groups of (w_pos * lpos + w_neg * lneg) / div  (one group unless per image: there the
  // sum of the images' losses = what the reference's loop adds to running_loss over the window)
  if ((int)threadIdx.x < hd.n) {
    const int h2 = threadIdx.x;
    float total = 0.f;
    for (int g2 = 0; g2 < G; ++g2) {
      float wp, wn, idv;
      cbce_weights(nm, sc.npos, g2, per_group, wp, wn, idv);
      const double lp = __hip_atomic_load(&sc.lpos[(size_t)h2 * G + g2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const double ln = __hip_atomic_load(&sc.lneg[(size_t)h2 * G + g2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float l = (float)(((double)wp * lp + (double)wn * ln) * (double)idv);
      total = g2 == 0 ? l : total + l;      // (fp32 adds in image order: what `running_loss += loss.item()` does on the host in double is within 1 ulp of this)
    }
    hd.loss[h2][0] = total;
    if (hd.running[h2] != nullptr) hd.running[h2][0] += total;      // running_loss += loss (train_online.py:128) without a host round trip or an extra launch
  }
  __syncthreads();
  // leave the scratch zero for the next call (everything this call accumulated has been consumed)
  for (int i = threadIdx.x; i < G; i += blockDim.x) sc.npos[i] = 0ull;
  for (int i = threadIdx.x; i < 2 * hd.n * G; i += blockDim.x) sc.lpos[i] = 0.0;      // (lpos and lneg are contiguous)
  if (threadIdx.x == 0) sc.ticket[0] = 0u;
}

__global__ void scale_kernel(const float* __restrict__ x, const float* __restrict__ scalar, float* __restrict__ y, long count) {
  const float s = scalar[0];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) y[i] = x[i] * s;
}

__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, long count,
                           float lr, float momentum, float wd, int first) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
    // roundings exactly as torch.optim.SGD's foreach kernels: d and p are fused multiply-adds (a + alpha*b), the momentum
    // buffer is buf.mul_(m) THEN .add_(d) -- two roundings (checked bit for bit against torch on the GPU)
#pragma clang fp contract(off)
    const float pv = p[i];
    const float d = __builtin_fmaf(wd, pv, g[i]);
    const float mb = momentum * buf[i];
    const float b = first ? d : mb + d;
    buf[i] = b;
    p[i] = __builtin_fmaf(-lr, b, pv);
  }
}

inline int grid_for(long total, int cap) {
  long b = (total + 255) / 256;
  return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" size_t osvos_cbce_scratch_bytes(int n_heads, int N, int flags) {
  const size_t G = (flags & OSVOS_CBCE_PER_IMAGE) ? (size_t)(N > 0 ? N : 1) : 1, H = (size_t)(n_heads > 0 ? n_heads : 1);
  return 8 * G + 16 * H * G + 8;
}

extern "C" int osvos_cbce_step_ex(const float* const* outs, const float* label, float* const* losses, float* const* grads, void* scratch, long count,
                                  int N, int mode, int flags, const float* counts, int n_heads, const float* grad_scales, float* const* running,
                                  void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OSVOS_ARG_CHECK(outs && label && losses && scratch && grad_scales && count > 0 && N > 0, "cbce: bad arguments");
  OSVOS_ARG_CHECK(n_heads >= 1 && n_heads <= kMaxHeads, "cbce: %d heads (1..%d)", n_heads, kMaxHeads);
  OSVOS_ARG_CHECK(mode >= 0 && mode <= 2, "cbce: mode %d", mode);
  const bool per_image = (flags & OSVOS_CBCE_PER_IMAGE) != 0;
  OSVOS_ARG_CHECK((flags & ~(OSVOS_CBCE_PER_IMAGE | OSVOS_CBCE_SCRATCH_ZEROED)) == 0 && !(per_image && counts != nullptr),
                  "cbce: flags 0x%x (per-image counts and external counts exclude each other)", flags);
  OSVOS_ARG_CHECK(!per_image || (count % N == 0 && N <= 65535), "cbce: per-image mode needs count %% N == 0 (%ld, %d)", count, N);
  const int G = per_image ? N : 1;
  const long per_group = count / G;
  CbceHeads hd;
  hd.n = n_heads;
  uintptr_t align = (uintptr_t)label;
  for (int h = 0; h < kMaxHeads; ++h) {
    const bool live = h < n_heads;
    OSVOS_ARG_CHECK(!live || (outs[h] && losses[h]), "cbce: head %d: null pointer", h);
    hd.out[h] = live ? outs[h] : nullptr;
    hd.grad[h] = live && grads ? grads[h] : nullptr;
    hd.loss[h] = live ? losses[h] : nullptr;
    hd.running[h] = live && running ? running[h] : nullptr;
    hd.gscale[h] = live ? grad_scales[h] : 0.f;
    if (live) align |= (uintptr_t)hd.out[h] | (uintptr_t)hd.grad[h];
  }
  // 16-byte sweeps need every group of every tensor aligned; anything else (an offset view, odd image sizes in the per-image mode) takes the
  // element-wise sweep -- slower, same numbers, not an error
  const bool vec = align % 16 == 0 && (G == 1 || per_group % 4 == 0);
  const long n4 = vec ? per_group >> 2 : 0;
  CbceNorm nm;
  nm.counts = counts; nm.mode = mode; nm.N = N; nm.per_image = per_image ? 1 : 0;
  const ScratchView sc = scratch_view(scratch, G, n_heads);
  if (!(flags & OSVOS_CBCE_SCRATCH_ZEROED)) OSVOS_HIP_CHECK(hipMemsetAsync(scratch, 0, osvos_cbce_scratch_bytes(n_heads, N, flags), stream));
  // one double atomic pair per workgroup: few workgroups for a single frame (11 us), more for batches (94 -> ~25 us at batch 12)
  const long work = vec ? n4 : per_group;
  int g = grid_for(work, count > (1L << 21) ? 512 : 128);
  if (G > 1) { g = (g + G - 1) / G; if (g < 16) g = 16; }
  if (counts == nullptr) hipLaunchKernelGGL(cbce_count_kernel, dim3(g, G), dim3(256), 0, stream, label, per_group, n4, sc.npos);
  hipLaunchKernelGGL(cbce_main_kernel, dim3(g, G, n_heads), dim3(256), 0, stream, hd, label, per_group, n4, nm, sc, G);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

extern "C" int osvos_cbce_step(const float* out, const float* label, float* loss, float* grad, void* scratch,
                               long count, int N, int mode, float grad_scale, float* running, void* stream_) {
  OSVOS_ARG_CHECK(out && label && loss && scratch, "cbce: bad arguments");
  const float* outs[1] = {out};
  float* losses[1] = {loss};
  float* grads[1] = {grad};
  float* runs[1] = {running};
  return osvos_cbce_step_ex(outs, label, losses, grads, scratch, count, N, mode, 0, nullptr, 1, &grad_scale, runs, stream_);
}

extern "C" int osvos_cbce(const float* out, const float* label, float* loss, float* grad, void* scratch,
                          long count, int N, int mode, void* stream_) {
  return osvos_cbce_step(out, label, loss, grad, scratch, count, N, mode, 1.f, nullptr, stream_);
}

extern "C" int osvos_cbce_step_multi(const float* const* outs, const float* label, float* const* losses, float* const* grads, void* scratch,
                                     long count, int N, int mode, int n_heads, const float* grad_scales, float* const* running, void* stream_) {
  return osvos_cbce_step_ex(outs, label, losses, grads, scratch, count, N, mode, 0, nullptr, n_heads, grad_scales, running, stream_);
}

extern "C" int osvos_scale(const float* x, const float* scalar, float* y, long count, void* stream) {
  OSVOS_ARG_CHECK(x && scalar && y && count > 0, "scale: bad arguments");
  hipLaunchKernelGGL(scale_kernel, dim3(grid_for(count, 2048)), dim3(256), 0, (hipStream_t)stream, x, scalar, y, count);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

// Every parameter tensor of the network in ONE launch: the tensor table rides in the kernel arguments (<= 64 entries),
// a workgroup walks the flat index space [0, total) in 1024-element blocks and finds its tensor by scanning the
// prefix sums.  Replaces torch.optim.SGD's per-group foreach kernels (train_online.py:79-88,147; train_parent.py:87-103,170).
struct SgdTable {
  float* p[OSVOS_SGD_MAX_TENSORS];
  const float* g[OSVOS_SGD_MAX_TENSORS];
  float* buf[OSVOS_SGD_MAX_TENSORS];
  long start[OSVOS_SGD_MAX_TENSORS + 1];      // in 1024-element blocks
  float lr[OSVOS_SGD_MAX_TENSORS], wd[OSVOS_SGD_MAX_TENSORS];
  long count[OSVOS_SGD_MAX_TENSORS];
  int n;
};

__global__ __launch_bounds__(256) void sgd_multi_kernel(SgdTable t, float momentum, int first) {
  for (long blk = blockIdx.x; blk < t.start[t.n]; blk += gridDim.x) {
    int k = 0;
    while (blk >= t.start[k + 1]) ++k;                       // uniform per workgroup: scalar loop
    const long base = (blk - t.start[k]) * 1024;
    float* __restrict__ p = t.p[k];
    const float* __restrict__ g = t.g[k];
    float* __restrict__ buf = t.buf[k];
    const float lr = t.lr[k], wd = t.wd[k];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long i = base + j * 256 + threadIdx.x;
      if (i < t.count[k]) {
#pragma clang fp contract(off)      // same roundings as sgd_kernel above
        const float pv = p[i];
        const float d = __builtin_fmaf(wd, pv, g[i]);
        const float mb = momentum * buf[i];
        const float b = first ? d : mb + d;
        buf[i] = b;
        p[i] = __builtin_fmaf(-lr, b, pv);
      }
    }
  }
}

extern "C" int osvos_sgd_step_multi(float* const* params, const float* const* grads, float* const* bufs, const long* counts,
                                    const float* lrs, const float* wds, int n, float momentum, int first, void* stream) {
  OSVOS_ARG_CHECK(params && grads && bufs && counts && lrs && wds && n >= 0, "sgd_step_multi: null table");
  for (int at = 0; at < n; at += OSVOS_SGD_MAX_TENSORS) {
    SgdTable t;
    t.n = n - at < OSVOS_SGD_MAX_TENSORS ? n - at : OSVOS_SGD_MAX_TENSORS;
    t.start[0] = 0;
    for (int k = 0; k < t.n; ++k) {
      OSVOS_ARG_CHECK(params[at + k] && grads[at + k] && bufs[at + k] && counts[at + k] > 0, "sgd_step_multi: tensor %d is null / empty", at + k);
      t.p[k] = params[at + k]; t.g[k] = grads[at + k]; t.buf[k] = bufs[at + k];
      t.count[k] = counts[at + k]; t.lr[k] = lrs[at + k]; t.wd[k] = wds[at + k];
      t.start[k + 1] = t.start[k] + (counts[at + k] + 1023) / 1024;
    }
    if (t.n == 0) break;
    const long blocks = t.start[t.n] < 4096 ? t.start[t.n] : 4096;
    hipLaunchKernelGGL(sgd_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, t, momentum, first);
    OSVOS_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int osvos_sgd_step(float* p, const float* g, float* buf, long count, float lr, float momentum,
                              float weight_decay, int first, void* stream) {
  OSVOS_ARG_CHECK(p && g && buf && count > 0, "sgd_step: bad arguments");
  hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(count, 4096)), dim3(256), 0, (hipStream_t)stream, p, g, buf, count, lr, momentum, weight_decay, first);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

// ---- result writer / evaluator (reference train_online.py:181-189: sigmoid -> scipy.misc.imsave, which min-max byte-scales;
//      the DAVIS region measure J = |P & G| / |P | G|, which the reference leaves to an external toolkit) ----------------------
namespace {

__device__ inline float sigmoidf_acc(float x) { return 1.f / (1.f + expf(-x)); }

// mm[2 n] / mm[2 n + 1]: bit patterns of min / max sigmoid of image n (positive floats order like unsigned integers)
__global__ void mask_minmax_kernel(const float* __restrict__ logits, long count, unsigned* __restrict__ mm) {
  const int n = blockIdx.y;
  const float* x = logits + (size_t)n * count;
  float lo = INFINITY, hi = -INFINITY;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
    const float v = x[i];
    lo = fminf(lo, v);
    hi = fmaxf(hi, v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, o, 64));
    hi = fmaxf(hi, __shfl_xor(hi, o, 64));
  }
  if ((threadIdx.x & 63) == 0 && lo <= hi) {     // sigmoid is monotone: min/max of the probabilities = sigmoid of min/max logit
    atomicMin(&mm[2 * n], __float_as_uint(sigmoidf_acc(lo)));
    atomicMax(&mm[2 * n + 1], __float_as_uint(sigmoidf_acc(hi)));
  }
}

// scipy<=1.1 bytescale(data, cmin=min, cmax=max): (p - cmin) * (255 / (cmax - cmin)) clipped to [0, 255], + 0.5, truncated
__global__ void mask_bytescale_kernel(const float* __restrict__ logits, long count, const unsigned* __restrict__ mm, unsigned char* __restrict__ out) {
  const int n = blockIdx.y;
  const float cmin = __uint_as_float(mm[2 * n]), cmax = __uint_as_float(mm[2 * n + 1]);
  float cscale = cmax - cmin;
  if (cscale == 0.f) cscale = 1.f;
  const float scale = (float)(255.0 / (double)cscale);
  const float* x = logits + (size_t)n * count;
  unsigned char* o = out + (size_t)n * count;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
    float b = (sigmoidf_acc(x[i]) - cmin) * scale;
    b = fminf(fmaxf(b, 0.f), 255.f) + 0.5f;
    o[i] = (unsigned char)b;
  }
}

// counts[2 n] = |P & G|, counts[2 n + 1] = |P | G| with P = logit > thr, G = gt > 0.5
__global__ void mask_iou_kernel(const float* __restrict__ logits, const float* __restrict__ gt, long count, float thr,
                                unsigned long long* __restrict__ counts) {
  const int n = blockIdx.y;
  const float* x = logits + (size_t)n * count;
  const float* g = gt + (size_t)n * count;
  unsigned inter = 0, uni = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
    const bool p = x[i] > thr, q = g[i] > 0.5f;
    inter += (p && q) ? 1u : 0u;
    uni += (p || q) ? 1u : 0u;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    inter += __shfl_xor(inter, o, 64);
    uni += __shfl_xor(uni, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&counts[2 * n], (unsigned long long)inter);
    atomicAdd(&counts[2 * n + 1], (unsigned long long)uni);
  }
}

}  // namespace

// logits fp32 [N][count] -> out uint8 [N][count]; scratch: 2 N unsigned (device)
extern "C" int osvos_mask_to_bytes(const float* logits, unsigned char* out, void* scratch, long count, int N, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OSVOS_ARG_CHECK(logits && out && scratch && count > 0 && N > 0, "mask_to_bytes: bad arguments");
  unsigned* mm = reinterpret_cast<unsigned*>(scratch);
  // min slots start at 0xffffffff, max slots at 0 : one memset pattern per half is not available, so two tiny fills
  OSVOS_HIP_CHECK(hipMemsetAsync(mm, 0, sizeof(unsigned) * 2 * N, stream));
  for (int n = 0; n < N; ++n) OSVOS_HIP_CHECK(hipMemsetAsync(mm + 2 * n, 0xff, sizeof(unsigned), stream));
  const dim3 grid((unsigned)grid_for(count, 1024), (unsigned)N);
  hipLaunchKernelGGL(mask_minmax_kernel, grid, dim3(256), 0, stream, logits, count, mm);
  OSVOS_LAUNCH_CHECK();
  hipLaunchKernelGGL(mask_bytescale_kernel, grid, dim3(256), 0, stream, logits, count, mm, out);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

// counts: 2 N unsigned long long (device), overwritten with {intersection, union} pixel counts per image
extern "C" int osvos_mask_iou_counts(const float* logits, const float* gt, void* counts, long count, int N, float logit_threshold, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OSVOS_ARG_CHECK(logits && gt && counts && count > 0 && N > 0, "mask_iou_counts: bad arguments");
  OSVOS_HIP_CHECK(hipMemsetAsync(counts, 0, sizeof(unsigned long long) * 2 * N, stream));
  hipLaunchKernelGGL(mask_iou_kernel, dim3((unsigned)grid_for(count, 1024), (unsigned)N), dim3(256), 0, stream, logits, gt, count, logit_threshold,
                     reinterpret_cast<unsigned long long*>(counts));
  OSVOS_LAUNCH_CHECK();
  return 0;
}
