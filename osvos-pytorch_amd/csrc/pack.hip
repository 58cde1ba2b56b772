// Layout conversions and weight packing (fp32 flavour; bf16 variants live next to their kernels).
//   NCHW <-> NHWC ................ boundary of OSVOS.forward (reference vgg_osvos.py:59, NCHW fp32)
//   OIHW -> MFMA operand packs ... nn.Conv2d.weight [Cout,Cin,3,3] (reference vgg_osvos.py:41,142)
#include "common.h"

namespace {

// One thread per PIXEL (block = 256 consecutive pixels of one image, blockIdx.y = image): the planes of the NCHW side are read / written as
// coalesced runs of consecutive pixels, the NHWC side as whole 16-byte channel groups.  (The first forms worked per ELEMENT off a flat 64-bit
// index: three divisions per element and plane-strided accesses inside a wave -- 108 us for the batch-12 input, 2.7 TB/s of 6.3.)
template <int CPAD>
__global__ __launch_bounds__(256) void nchw_to_nhwc_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, bf16_t* __restrict__ dstbf,
                                                               int C, unsigned hw) {
  const unsigned p = blockIdx.x * 256u + threadIdx.x;
  if (p >= hw) return;
  const size_t n = blockIdx.y;
  float v[CPAD];
#pragma unroll
  for (int c = 0; c < CPAD; ++c) v[c] = c < C ? src[(n * C + c) * hw + p] : 0.f;
  f32x4* d4 = reinterpret_cast<f32x4*>(dst + (n * hw + p) * CPAD);
#pragma unroll
  for (int q = 0; q < CPAD / 4; ++q) d4[q] = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
  if (dstbf != nullptr) {
    struct HB { bf16_t h[CPAD]; } hb;      // one 16- / 8-byte store
#pragma unroll
    for (int c = 0; c < CPAD; ++c) hb.h[c] = f32_to_bf16(v[c]);
    *reinterpret_cast<HB*>(dstbf + (n * hw + p) * CPAD) = hb;
  }
}

// generic channel padding (not 4 / 8): per element
__global__ void nchw_to_nhwc_f32_any_kernel(const float* __restrict__ src, float* __restrict__ dst, bf16_t* __restrict__ dstbf,
                                            int N, int C, int H, int W, int cpad) {
  const long total = (long)N * H * W * cpad;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cpad);
    const long pix = i / cpad;
    const long hw = (long)H * W;
    const long n = pix / hw, p = pix % hw;
    const float v = c < C ? src[(n * C + c) * hw + p] : 0.f;
    dst[i] = v;
    if (dstbf != nullptr) dstbf[i] = f32_to_bf16(v);
  }
}

// cs == 4 (the 3-channel input gradient): one 16-byte read per pixel, C coalesced plane writes
__global__ __launch_bounds__(256) void nhwc4_to_nchw_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, unsigned hw) {
  const unsigned p = blockIdx.x * 256u + threadIdx.x;
  if (p >= hw) return;
  const size_t n = blockIdx.y;
  const f32x4 v = reinterpret_cast<const f32x4*>(src)[n * hw + p];
#pragma unroll
  for (int c = 0; c < 4; ++c)
    if (c < C) dst[(n * C + c) * hw + p] = v[c];
}

__global__ void nhwc_to_nchw_f32_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                        int N, int C, int H, int W, int cs) {
  const long hw = (long)H * W;
  const long total = (long)N * C * hw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long p = i % hw;
    const long nc = i / hw;
    const long n = nc / C, c = nc % C;
    dst[i] = src[(n * hw + p) * cs + c];
  }
}

// wpk[((tap*CQ + cq)*CoutP + co)*4 + e] = W[co][4cq+e][tap]   (zero padded)
__global__ void pack_fwd_f32_kernel(const float* __restrict__ w, float* __restrict__ wpk,
                                    int Cout, int Cin, int CinP, int CoutP) {
  const int CQ = CinP / 4;
  const long total = 9L * CQ * CoutP * 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 3);
    long t = i >> 2;
    const int co = (int)(t % CoutP);
    t /= CoutP;
    const int cq = (int)(t % CQ);
    const int tap = (int)(t / CQ);
    const int ci = cq * 4 + e;
    wpk[i] = (co < Cout && ci < Cin) ? w[((long)co * Cin + ci) * 9 + tap] : 0.f;
  }
}

// data-gradient pack: dX = conv3x3(dY, Wd), Wd[ci][co][r'][s'] = W[co][ci][2-r'][2-s']
// wpk[((tap'*CQo + coq)*CinP + ci)*4 + e] = W[4coq+e][ci][8 - tap']
__global__ void pack_dgrad_f32_kernel(const float* __restrict__ w, float* __restrict__ wpk,
                                      int Cout, int Cin, int CoutK, int CinP) {
  const int CQ = CoutK / 4;
  const long total = 9L * CQ * CinP * 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 3);
    long t = i >> 2;
    const int ci = (int)(t % CinP);
    t /= CinP;
    const int coq = (int)(t % CQ);
    const int tap = (int)(t / CQ);
    const int co = coq * 4 + e;
    wpk[i] = (co < Cout && ci < Cin) ? w[((long)co * Cin + ci) * 9 + (8 - tap)] : 0.f;
  }
}

inline int grid_for(long total) {
  long b = (total + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

int osvos_nchw_to_nhwc_f32(const float* src, float* dst, void* dstbf, int N, int C, int H, int W, int cpad, hipStream_t stream) {
  OSVOS_ARG_CHECK(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && cpad >= C, "nchw_to_nhwc: bad arguments");
  const long hw = (long)H * W;
  if ((cpad == 8 || cpad == 4) && hw < (1L << 31) && N <= 65535) {
    const dim3 grid((unsigned)((hw + 255) / 256), (unsigned)N);
    if (cpad == 8) hipLaunchKernelGGL(nchw_to_nhwc_f32_kernel<8>, grid, dim3(256), 0, stream, src, dst, reinterpret_cast<bf16_t*>(dstbf), C, (unsigned)hw);
    else hipLaunchKernelGGL(nchw_to_nhwc_f32_kernel<4>, grid, dim3(256), 0, stream, src, dst, reinterpret_cast<bf16_t*>(dstbf), C, (unsigned)hw);
  } else {
    hipLaunchKernelGGL(nchw_to_nhwc_f32_any_kernel, dim3(grid_for((long)N * H * W * cpad)), dim3(256), 0, stream, src, dst, reinterpret_cast<bf16_t*>(dstbf), N, C, H, W, cpad);
  }
  OSVOS_LAUNCH_CHECK();
  return 0;
}

int osvos_nhwc_to_nchw_f32(const float* src, float* dst, int N, int C, int H, int W, int cs, hipStream_t stream) {
  OSVOS_ARG_CHECK(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && cs >= C, "nhwc_to_nchw: bad arguments");
  if (cs == 4 && C <= 4 && (long)H * W < (1L << 31) && N <= 65535)
    hipLaunchKernelGGL(nhwc4_to_nchw_f32_kernel, dim3((unsigned)(((long)H * W + 255) / 256), (unsigned)N), dim3(256), 0, stream, src, dst, C, (unsigned)((long)H * W));
  else
    hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel, dim3(grid_for((long)N * H * W * C)), dim3(256), 0, stream, src, dst, N, C, H, W, cs);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

int osvos_pack_fwd_f32(const float* w, float* wpk, int Cout, int Cin, hipStream_t stream) {
  OSVOS_ARG_CHECK(w && wpk && Cout > 0 && Cin > 0, "pack_fwd: bad arguments");
  const int CinP = osvos_cin_pad(Cin, OSVOS_F32), CoutP = osvos_cout_pad(Cout);
  hipLaunchKernelGGL(pack_fwd_f32_kernel, dim3(grid_for(9L * CinP * CoutP)), dim3(256), 0, stream, w, wpk, Cout, Cin, CinP, CoutP);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

int osvos_pack_dgrad_f32(const float* w, float* wpk, int Cout, int Cin, hipStream_t stream) {
  OSVOS_ARG_CHECK(w && wpk && Cout > 0 && Cin > 0, "pack_dgrad: bad arguments");
  const int CoutK = osvos_cin_pad(Cout, OSVOS_F32), CinP = osvos_cout_pad(Cin);
  hipLaunchKernelGGL(pack_dgrad_f32_kernel, dim3(grid_for(9L * CoutK * CinP)), dim3(256), 0, stream, w, wpk, Cout, Cin, CoutK, CinP);
  OSVOS_LAUNCH_CHECK();
  return 0;
}
