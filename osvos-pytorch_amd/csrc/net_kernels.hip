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
  const double* acc;      // 4 x {dwf[16], dwd[16], dbd, spare} then d(fuse.bias)
  float* fuse_w;
  float* fuse_b;
  float* dsn_w[4];
  float* dsn_b[4];
  int accumulate;
};

__global__ void head_grads_finalize_kernel(FinalizeArgs a) {
  const int t = threadIdx.x;   // 128 threads
  if (t < 64) {
    const int i = t >> 4, c = t & 15;
    const float v = (float)a.acc[34 * i + c];
    if (a.fuse_w) a.fuse_w[t] = a.accumulate ? a.fuse_w[t] + v : v;
  } else {
    const int u = t - 64, i = u >> 4, c = u & 15;
    const float v = (float)a.acc[34 * i + 16 + c];
    if (a.dsn_w[i]) a.dsn_w[i][c] = a.accumulate ? a.dsn_w[i][c] + v : v;
    if (c == 0 && a.dsn_b[i]) {
      const float b = (float)a.acc[34 * i + 32];
      a.dsn_b[i][0] = a.accumulate ? a.dsn_b[i][0] + b : b;
    }
  }
  if (t == 0 && a.fuse_b) {
    const float b = (float)a.acc[4 * 34];
    a.fuse_b[0] = a.accumulate ? a.fuse_b[0] + b : b;
  }
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
int osvos_head_grads_finalize(const double* acc, float* const* grads, int accumulate, int have_side, hipStream_t stream) {
  FinalizeArgs a;
  a.acc = acc;
  a.fuse_w = grads[50];
  a.fuse_b = grads[51];
  for (int i = 0; i < 4; ++i) {
    a.dsn_w[i] = have_side ? grads[42 + 2 * i] : nullptr;
    a.dsn_b[i] = have_side ? grads[43 + 2 * i] : nullptr;
  }
  a.accumulate = accumulate;
  hipLaunchKernelGGL(head_grads_finalize_kernel, dim3(1), dim3(128), 0, stream, a);
  OSVOS_LAUNCH_CHECK();
  return 0;
}
