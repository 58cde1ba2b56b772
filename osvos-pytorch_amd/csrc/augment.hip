// Training-time input pipeline on the device.  Replaces, for one frame,
//   dataloaders/davis_2016.py:99-106   BGR uint8 -> float32 minus mean; label / max(label.max(), 1e-8)
//   custom_transforms.py:87-99         RandomHorizontalFlip (cv2.flip(.., 1))
//   custom_transforms.py:21-52         ScaleNRotate: cv2.warpAffine(INTER_CUBIC image, INTER_NEAREST 0/1 mask), border 0
//   custom_transforms.py:102-121       ToTensor (HWC -> CHW)
// in one pass over the output pixels: the uint8 frame crosses PCIe once, the float32 NCHW tensors the network consumes are
// produced in HBM.  The warp follows OpenCV's algorithm (imgwarp.cpp: 10-bit fixed-point coordinates, 5-bit interpolation
// table index, interpolateCubic with A = -0.75, row-wise / tap-wise float32 summation order of remapBicubic) -- restated, with
// the same operation order, in oracle/augment_ref.py; OpenCV itself is absent here, so the restatement is unpinned.
#include "common.h"

namespace {

struct AugArgs {
  const unsigned char* img;      // [H][W][3] BGR
  const unsigned char* label;    // [H][W] or NULL
  float mean[3];
  double M[6];                   // dst -> src affine (cv::warpAffine's inverted matrix)
  int flip, warp, H, W;
  float* out_img;                // [3][H][W]
  float* out_gt;                 // [1][H][W]
  const unsigned* stats;         // [0] label max, [1] 256 - smallest non-zero label value (0: none); soft mask (-> cubic) iff they name different values
};

// One sweep over the label: stats[0] = its maximum, stats[1] = 256 - (its smallest NON-ZERO value) (0 for an all-zero label): the mask
// holds a value that is neither 0 nor the maximum -- the reference's ((gt == 0) | (gt == 1)).all() test on the normalised mask fails, the warp
// goes cubic -- exactly when smallest non-zero != maximum.  16-byte loads where the pointer allows, one pair of atomics per workgroup
// (round 3's pair of kernels issued 4096 same-address atomics and a second sweep: 49 + 6 us per frame next to a 37 us warp).
__global__ __launch_bounds__(256) void label_stats_kernel(const unsigned char* __restrict__ label, long count, unsigned* stats) {
  unsigned m = 0, inv = 0;      // inv = max over non-zero v of (256 - v)
  auto take = [&](unsigned v) { m = max(m, v); inv = max(inv, v != 0u ? 256u - v : 0u); };
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long)gridDim.x * blockDim.x;
  const long head = min(count, (long)((16 - (reinterpret_cast<size_t>(label) & 15)) & 15));      // bytes before the first 16-byte boundary
  const long n16 = (count - head) / 16;
  for (long i = tid; i < head; i += nth) take(label[i]);
  const uint4* p = reinterpret_cast<const uint4*>(label + head);
  for (long i = tid; i < n16; i += nth) {
    const uint4 q = p[i];
    const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int b = 0; b < 4; ++b) take((w[k] >> (8 * b)) & 0xffu);
  }
  for (long i = head + n16 * 16 + tid; i < count; i += nth) take(label[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    inv = max(inv, (unsigned)__shfl_xor((int)inv, o, 64));
  }
  __shared__ unsigned red[2][4];
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = m; red[1][threadIdx.x >> 6] = inv; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicMax(&stats[0], max(max(red[0][0], red[0][1]), max(red[0][2], red[0][3])));
    atomicMax(&stats[1], max(max(red[1][0], red[1][1]), max(red[1][2], red[1][3])));
  }
}

__device__ inline void cubic_coeffs(int fi, float* c) {
#pragma clang fp contract(off)
  const float x = (float)fi * (1.0f / 32.0f);
  const float A = -0.75f;
  c[0] = ((A * (x + 1.f) - 5.f * A) * (x + 1.f) + 8.f * A) * (x + 1.f) - 4.f * A;
  c[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  c[2] = ((A + 2.f) * (1.f - x) - (A + 3.f)) * (1.f - x) * (1.f - x) + 1.f;
  c[3] = 1.f - c[0] - c[1] - c[2];
}

__device__ inline int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

__global__ __launch_bounds__(256) void augment_kernel(AugArgs a) {
#pragma clang fp contract(off)
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= a.W) return;
  const int H = a.H, W = a.W;
  const long hw = (long)H * W;
  const float den = fmaxf((float)a.stats[0], 1e-8f);
  const bool gt_cubic = a.label != nullptr && a.stats[1] != 0u && 256u - a.stats[1] != a.stats[0];      // ((gt == 0) | (gt == 1)).all() is false
  auto pix = [&](int yy, int xx, int c) -> float {                    // the (flipped) mean-subtracted frame
    const int sx = a.flip ? W - 1 - xx : xx;
    return (float)a.img[((long)yy * W + sx) * 3 + c] - a.mean[c];
  };
  auto gtv = [&](int yy, int xx) -> float {
    if (a.label == nullptr) return 0.f;
    const int sx = a.flip ? W - 1 - xx : xx;
    return (float)a.label[(long)yy * W + sx] / den;
  };
  const long o = (long)y * W + x;
  if (!a.warp) {
#pragma unroll
    for (int c = 0; c < 3; ++c) a.out_img[c * hw + o] = pix(y, x, c);
    a.out_gt[o] = gtv(y, x);
    return;
  }
  // fixed-point source coordinates exactly as WarpAffineInvoker forms them (doubles, round half to even)
  const int adelta = __double2int_rn(a.M[0] * (double)x * 1024.0), bdelta = __double2int_rn(a.M[3] * (double)x * 1024.0);
  const int Xb = __double2int_rn((a.M[1] * (double)y + a.M[2]) * 1024.0), Yb = __double2int_rn((a.M[4] * (double)y + a.M[5]) * 1024.0);
  // ---- bicubic (image, and soft masks) ----
  const int Xc = (Xb + 16 + adelta) >> 5, Yc = (Yb + 16 + bdelta) >> 5;
  const int sx = sat_short(Xc >> 5) - 1, sy = sat_short(Yc >> 5) - 1;
  float tx[4], ty[4];
  cubic_coeffs(Xc & 31, tx);
  cubic_coeffs(Yc & 31, ty);
  const bool inside = sx >= 0 && sx < max(W - 3, 0) && sy >= 0 && sy < max(H - 3, 0);
  const bool outside = sx >= W || sx + 4 <= 0 || sy >= H || sy + 4 <= 0;
  auto cubic = [&](auto fetch) -> float {
    if (outside) return 0.f;
    if (inside) {
      float acc = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float row = ((fetch(sy + r, sx) * (ty[r] * tx[0]) + fetch(sy + r, sx + 1) * (ty[r] * tx[1])) + fetch(sy + r, sx + 2) * (ty[r] * tx[2])) +
                          fetch(sy + r, sx + 3) * (ty[r] * tx[3]);
        acc = r == 0 ? row : acc + row;
      }
      return acc;
    }
    float acc = 0.f;
    for (int i = 0; i < 4; ++i) {
      const int yi = sy + i;
      if (yi < 0 || yi >= H) continue;
      for (int j = 0; j < 4; ++j) {
        const int xj = sx + j;
        if (xj >= 0 && xj < W) acc = acc + fetch(yi, xj) * (ty[i] * tx[j]);
      }
    }
    return acc;
  };
#pragma unroll
  for (int c = 0; c < 3; ++c) a.out_img[c * hw + o] = cubic([&](int yy, int xx) { return pix(yy, xx, c); });
  if (gt_cubic) {
    a.out_gt[o] = cubic([&](int yy, int xx) { return gtv(yy, xx); });
  } else {      // nearest: round_delta = AB_SCALE / 2
    const int Xn = sat_short((Xb + 512 + adelta) >> 10), Yn = sat_short((Yb + 512 + bdelta) >> 10);
    a.out_gt[o] = (Xn >= 0 && Xn < W && Yn >= 0 && Yn < H) ? gtv(Yn, Xn) : 0.f;
  }
}

}  // namespace

// img: uint8 [H][W][3] (BGR, as cv2.imread returns it), label: uint8 [H][W] or NULL (device pointers).  Minv: HOST pointer to the 6
// doubles of the dst -> src affine (cv::warpAffine's inverted matrix) or NULL for no warp.  scratch: 2 unsigned (device).
extern "C" int osvos_augment_frame(const unsigned char* img, const unsigned char* label, const float* mean3, int flip, const double* Minv,
                                   float* out_img, float* out_gt, void* scratch, int H, int W, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OSVOS_ARG_CHECK(img && mean3 && out_img && out_gt && scratch && H > 0 && W > 0, "augment_frame: bad arguments");
  OSVOS_ARG_CHECK(H < 32768 && W < 32768, "augment_frame: frame larger than OpenCV's 16-bit coordinate range");
  unsigned* stats = reinterpret_cast<unsigned*>(scratch);
  OSVOS_HIP_CHECK(hipMemsetAsync(stats, 0, 2 * sizeof(unsigned), stream));
  if (label != nullptr) {
    const long count = (long)H * W;
    long b = (count / 16 + 255) / 256;
    b = b < 1 ? 1 : (b > 256 ? 256 : b);
    hipLaunchKernelGGL(label_stats_kernel, dim3((unsigned)b), dim3(256), 0, stream, label, count, stats);
    OSVOS_LAUNCH_CHECK();
  }
  AugArgs a;
  a.img = img; a.label = label; a.flip = flip ? 1 : 0; a.H = H; a.W = W; a.out_img = out_img; a.out_gt = out_gt; a.stats = stats;
  for (int c = 0; c < 3; ++c) a.mean[c] = mean3[c];
  a.warp = Minv != nullptr ? 1 : 0;
  for (int k = 0; k < 6; ++k) a.M[k] = Minv ? Minv[k] : 0.0;
  hipLaunchKernelGGL(augment_kernel, dim3((unsigned)((W + 255) / 256), (unsigned)H), dim3(256), 0, stream, a);
  OSVOS_LAUNCH_CHECK();
  return 0;
}
