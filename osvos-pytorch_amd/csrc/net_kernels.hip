// Small device helpers of the whole-network orchestration (net.cpp).
#include "common.h"

namespace {

constexpr int kMaxGather = 64;
struct GatherArgs {
  const float* src[kMaxGather];
  unsigned long long dst_off[kMaxGather];
  int count[kMaxGather];
  char* base;
};

// one workgroup per entry: copies biases / head weights / deconv filters into wbuf
__global__ void gather_small_kernel(GatherArgs a) {
  const int e = blockIdx.x;
  float* dst = reinterpret_cast<float*>(a.base + a.dst_off[e]);
  for (int i = threadIdx.x; i < a.count[e]; i += blockDim.x) dst[i] = a.src[e][i];
}

struct FinalizeArgs {
  const double* part[4];   // per scale: [nblk[i]][34] partials {dwf[16], dwd[16], dbd, spare}
  const double* fb_part;   // [fb_nblk] partial sums of dfused
  int nblk[4], fb_nblk;
  float* fuse_w;
  float* fuse_b;
  float* dsn_w[4];
  float* dsn_b[4];
  int accumulate;
};

// one wave per output value: lanes stride over the per-workgroup partials, shuffle-reduce
__global__ __launch_bounds__(64) void head_grads_finalize_kernel(FinalizeArgs a) {
  const int t = blockIdx.x, lane = threadIdx.x;   // 133 outputs: 64 fuse.weight, 64 score_dsn.weight, 4 score_dsn.bias, fuse.bias
  const double* src;
  int n, stride, col;
  float* dst;
  if (t < 128) {
    const int u = t & 63, i = u >> 4, c = u & 15;
    src = a.part[i]; n = a.nblk[i]; stride = 34; col = (t < 64 ? 0 : 16) + c;
    dst = t < 64 ? (a.fuse_w ? a.fuse_w + u : nullptr) : (a.dsn_w[i] ? a.dsn_w[i] + c : nullptr);
  } else if (t < 132) {
    const int i = t - 128;
    src = a.part[i]; n = a.nblk[i]; stride = 34; col = 32;
    dst = a.dsn_b[i];
  } else {
    src = a.fb_part; n = a.fb_nblk; stride = 1; col = 0;
    dst = a.fuse_b;
  }
  if (dst == nullptr) return;
  double s = 0.0;
  for (int b = lane; b < n; b += 64) s += src[(size_t)b * stride + col];
  s = wave_sum(s);
  if (lane == 0) *dst = a.accumulate ? *dst + (float)s : (float)s;
}

}  // namespace

int osvos_gather_small(const float* const* srcs, const size_t* dst_off, const int* counts, int n, void* wbuf, hipStream_t stream) {
  OSVOS_ARG_CHECK(n > 0 && n <= kMaxGather, "gather_small: %d entries", n);
  GatherArgs a;
  for (int i = 0; i < n; ++i) { a.src[i] = srcs[i]; a.dst_off[i] = dst_off[i]; a.count[i] = counts[i]; }
  a.base = reinterpret_cast<char*>(wbuf);
  hipLaunchKernelGGL(gather_small_kernel, dim3(n), dim3(256), 0, stream, a);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

// grads: the 52-entry state_dict-order array of osvos_net_backward (NULL entries skipped)
int osvos_head_grads_finalize(const double* const* part, const int* nblk, const double* fb_part, int fb_nblk,
                              float* const* grads, int accumulate, int have_side, hipStream_t stream) {
  FinalizeArgs a;
  for (int i = 0; i < 4; ++i) {
    a.part[i] = part[i];
    a.nblk[i] = nblk[i];
    a.dsn_w[i] = have_side ? grads[42 + 2 * i] : nullptr;
    a.dsn_b[i] = have_side ? grads[43 + 2 * i] : nullptr;
  }
  a.fb_part = fb_part;
  a.fb_nblk = fb_nblk;
  a.fuse_w = grads[50];
  a.fuse_b = grads[51];
  a.accumulate = accumulate;
  hipLaunchKernelGGL(head_grads_finalize_kernel, dim3(133), dim3(64), 0, stream, a);
  OSVOS_LAUNCH_CHECK();
  return 0;
}
