// 3x3 stride-1 pad-1 convolution, NHWC fp32, implicit GEMM on v_mfma_f32_32x32x2_f32 (gfx950).
//
// Replaces aten::convolution for nn.Conv2d(k=3, padding=1) (reference vgg_osvos.py:41,142-143)
// and, fed with the rotated/transposed weight pack, the data-gradient half of
// aten::convolution_backward.  Exact fp32: the MFMA is bitwise a k-ordered fmaf chain.
//
// GEMM view: M = output pixels, N = Cout, K = 9 * Cin.
//   * a workgroup (4 waves) owns a TW x TH spatial patch x BN output channels
//   * per 8-channel K chunk the (TH+2) x (TW+2) input halo is staged ONCE into LDS as two
//     planes of 16-byte channel quads ([quad][row][col] float4) and reused by all 9 taps:
//     a tap is just an LDS address offset (r*PITCH + s), so the 9x im2col blow-up never
//     touches L2/HBM
//   * MFMA row block = 32 pixels laid out RBW wide x 32/RBW tall; lane l (pixel l&31, half l>>5)
//     reads ONE ds_read_b128 = channels 4*(l>>5)..+3 of its pixel and feeds 4 MFMAs
//     (k = l>>5 selects the quad) -> conflict-free 16-B slots, 1 LDS instr per 4 MFMAs
//   * weights are pre-packed [tap][Cin/4][CoutP][4] so the B operand is the same one-b128 pattern
//   * global->LDS staging is register double-buffered: loads of chunk k+1 are issued before the
//     MFMAs of chunk k and written to the other LDS buffer after them (one barrier per chunk)
//   * blockIdx -> tile mapping puts the Cout tile in the low bits so that each XCD (block b runs
//     on XCD b % 8) keeps re-using the same weight slice from its private L2
#include "common.h"
#include "kernels.h"

namespace {

struct ConvArgs {
  const float* x;
  const float* wpk;
  const float* bias;
  const float* mask;
  float* y;
  int N, H, W, Cin, Cout, CoutP, y_cs;
  int tiles_x, tiles_y, nct;
  int relu, map, nsp;
  int ksplit;         // > 1: blockIdx.y = K part; raw partial sums go to `part`, epilogue runs in conv_splitk_finalize
  float* part;        // [ksplit][N][H][W][Cout]
  unsigned long long* prof;   // phase cycle counters (-DOSVOS_CONV_PROF builds only; tools/conv_phase_probe.py)
};

constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
constexpr int pitch_for(int rbw, int hw) {
  // RBW 32: any pitch is conflict free; RBW 16: pitch % 16 == 0; RBW 8: pitch % 16 == 8
  return rbw == 32 ? hw : (rbw == 16 ? cdiv(hw, 16) * 16 : cdiv(hw - 8, 16) * 16 + 8);
}

template <int RBW_, int TBX_, int TBY_, int NB_, int WGM_, int WGN_>
struct Cfg {
  static constexpr int RBW = RBW_, TBX = TBX_, TBY = TBY_, NB = NB_, WGM = WGM_, WGN = WGN_;
  static constexpr int RBH = 32 / RBW;
  static constexpr int TW = TBX * RBW, TH = TBY * RBH;
  static constexpr int HWD = TW + 2, HHT = TH + 2;
  static constexpr int PITCH = pitch_for(RBW, HWD);
  static constexpr int PLANE = HHT * PITCH;
  static constexpr int BN = NB * 32;
  static constexpr int A_F4 = 2 * PLANE;
  static constexpr int B_F4 = 9 * 2 * BN;
  static constexpr int BUF_F4 = A_F4 + B_F4;
  static constexpr int A_LOAD = HHT * HWD * 2;
  static constexpr int NA = cdiv(A_LOAD, 256);
  static constexpr int NBL = cdiv(B_F4, 256);
  static constexpr int MB = TBX * TBY;
  static constexpr int WM = MB / WGM, WN = NB / WGN;
  static constexpr size_t LDS_BYTES = (size_t)2 * (BUF_F4 + 1) * 16;   // two buffers, each with one spare slot
  static_assert(WGM * WGN == 4, "4 waves per workgroup");
  static_assert(MB % WGM == 0 && NB % WGN == 0, "wave grid must divide the tile");
};

template <class C>
__global__ __launch_bounds__(256) void conv3x3_f32_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* lds = reinterpret_cast<f32x4*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave / C::WGN, wn = wave % C::WGN;

  // blockIdx -> (Cout tile, spatial tile).  map 0: Cout tile in the low bits, so XCD b % 8 keeps
  // one weight slice hot in its L2 (deep layers: weights >> activations).  map 1: spatial tile
  // = 8 * (b / (8 nct)) + b % 8, Cout tiles consecutive on the SAME XCD, so the input halo tile is
  // fetched from HBM once per XCD instead of once per Cout tile (shallow layers).
  int sp, ct;
  if (a.map == 0) {
    sp = blockIdx.x / a.nct;
    ct = blockIdx.x % a.nct;
  } else {
    const int j = blockIdx.x >> 3;
    ct = j % a.nct;
    sp = (j / a.nct) * 8 + (blockIdx.x & 7);
    if (sp >= a.nsp) return;
  }
  const int tx = sp % a.tiles_x;
  sp /= a.tiles_x;
  const int ty = sp % a.tiles_y;
  const int n = sp / a.tiles_y;
  const int x0 = tx * C::TW, y0 = ty * C::TH, co0 = ct * C::BN;
  const int CQ = a.Cin >> 2;

  const float* ximg = a.x + (size_t)n * a.H * a.W * a.Cin;

  // ---- per-thread staging descriptors (invariant over K chunks) --------------------------
  // Staging uses raw buffer loads: a lane whose halo pixel is outside the image (or that has no slot) carries byte
  // offset 0x80000000 -- past num_records -- and the hardware returns zeros.  No per-load branch is left in the K loop
  // (hipcc turns `if (ok) v = *p` into an exec-mask branch pair per load); the chunk advance rides in the scalar offset.
  constexpr unsigned OOB = 0x80000000u;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ximg), 0, (int)((size_t)a.H * a.W * a.Cin * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wpk), 0, (int)((size_t)9 * CQ * a.CoutP * 16), 0x00020000);
  unsigned a_off[C::NA];
  int a_dst[C::NA];
#pragma unroll
  for (int i = 0; i < C::NA; ++i) {
    const int e = tid + i * 256;
    const int h = e & 1, pix = e >> 1;
    const int hy = pix / C::HWD, hx = pix % C::HWD;
    const int gy = y0 + hy - 1, gx = x0 + hx - 1;
    const bool slot = e < C::A_LOAD;
    a_dst[i] = slot ? h * C::PLANE + hy * C::PITCH + hx : C::BUF_F4;       // lanes without a slot write the spare slot of their buffer
    a_off[i] = (slot && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? (unsigned)(((gy * a.W + gx) * a.Cin + 4 * h) * 4) : OOB;
  }
  unsigned b_off[C::NBL];
#pragma unroll
  for (int i = 0; i < C::NBL; ++i) {
    const int e = tid + i * 256;
    const int tap = e / (2 * C::BN), rem = e % (2 * C::BN);
    const int h = rem / C::BN, nn = rem % C::BN;
    b_off[i] = (e < C::B_F4 && co0 + nn < a.CoutP) ? (unsigned)(((tap * CQ + h) * a.CoutP + co0 + nn) * 16) : OOB;
  }

  u32x4 ra[C::NA], rb[C::NBL];
  auto load_chunk = [&](int kc) {
#pragma unroll
    for (int i = 0; i < C::NA; ++i) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, a_off[i], kc * 32, 0);
#pragma unroll
    for (int i = 0; i < C::NBL; ++i) rb[i] = __builtin_amdgcn_raw_buffer_load_b128(wrs, b_off[i], kc * 2 * a.CoutP * 16, 0);
  };
  auto store_chunk = [&](int buf) {
    f32x4* As = lds + buf * (C::BUF_F4 + 1);
    f32x4* Bs = As + C::A_F4;
#pragma unroll
    for (int i = 0; i < C::NA; ++i) As[a_dst[i]] = __builtin_bit_cast(f32x4, ra[i]);
#pragma unroll
    for (int i = 0; i < C::NBL; ++i)
      if (C::B_F4 % 256 == 0 || tid + i * 256 < C::B_F4) Bs[tid + i * 256] = __builtin_bit_cast(f32x4, rb[i]);
  };

  // ---- per-lane fragment addresses ---------------------------------------------------------
  int a_idx[C::WM];
#pragma unroll
  for (int mi = 0; mi < C::WM; ++mi) {
    const int mb = wm * C::WM + mi;
    const int mbx = mb % C::TBX, mby = mb / C::TBX;
    const int dy = li / C::RBW, dx = li % C::RBW;
    a_idx[mi] = lh * C::PLANE + (mby * C::RBH + dy) * C::PITCH + mbx * C::RBW + dx;
  }
  const int b_idx = lh * C::BN + wn * C::WN * 32 + li;

  f32x16 acc[C::WM][C::WN];
#pragma unroll
  for (int mi = 0; mi < C::WM; ++mi)
#pragma unroll
    for (int ni = 0; ni < C::WN; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int nch_all = a.Cin >> 3;
  const int kc_begin = (int)((long)nch_all * blockIdx.y / a.ksplit);
  const int nchunks = (int)((long)nch_all * (blockIdx.y + 1) / a.ksplit);      // end (exclusive) of this K part
#ifdef OSVOS_CONV_PROF
  unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tq, tp = __builtin_amdgcn_s_memtime();
  const unsigned long long t_begin = tp;
#define PROF_MARK(k) do { tq = __builtin_amdgcn_s_memtime(); pt[k] += tq - tp; tp = tq; } while (0)
#else
#define PROF_MARK(k) do { } while (0)
#endif
  load_chunk(kc_begin);
  store_chunk(kc_begin & 1);
  __syncthreads();
  PROF_MARK(0);
  for (int kc = kc_begin; kc < nchunks; ++kc) {
    const bool more = kc + 1 < nchunks;
    if (more) load_chunk(kc + 1);
    PROF_MARK(5);
    const f32x4* As = lds + (kc & 1) * (C::BUF_F4 + 1);
    const f32x4* Bs = As + C::A_F4;
    // taps are software-pipelined: the fragments of tap+1 are requested before the MFMAs of tap issue
    f32x4 fa[2][C::WM], fb[2][C::WN];
    auto ldfrag = [&](int tap, int set) {
      const int r = tap / 3, s = tap % 3;
#pragma unroll
      for (int mi = 0; mi < C::WM; ++mi) fa[set][mi] = As[a_idx[mi] + r * C::PITCH + s];
#pragma unroll
      for (int ni = 0; ni < C::WN; ++ni) fb[set][ni] = Bs[b_idx + tap * 2 * C::BN + ni * 32];
    };
    ldfrag(0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      // sched_barrier pins "request tap+1, then multiply tap": hipcc otherwise sinks the ds_reads to
      // their first use and every tap exposes an LDS round trip to the matrix pipe
      if (tap + 1 < 9) ldfrag(tap + 1, (tap + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < C::WM; ++mi)
#pragma unroll
          for (int ni = 0; ni < C::WN; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[tap & 1][ni][j], fa[tap & 1][mi][j], acc[mi][ni], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    PROF_MARK(6);
#ifdef OSVOS_CONV_PROF
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    PROF_MARK(2);
#endif
    if (more) store_chunk((kc + 1) & 1);
    PROF_MARK(3);
    __syncthreads();
    PROF_MARK(1);
  }

  // ---- epilogue.  The weight fragment is the FIRST MFMA operand, so D = [cout rows][pixel columns]: lane (li, lh) holds
  // pixel li of its M block and, per accumulator, couts 8 q + 4 lh + (0..3) in registers 4q..4q+3 -- four consecutive
  // NHWC channels = one 16-byte store.  Stores (and bias / mask loads) are raw buffer accesses whose offset is pushed
  // out of range for pixels outside the image and couts past Cout: no branches, no 64-bit address arithmetic (the
  // former per-element dword epilogue cost a wave 500 cycles per store instruction, tools/conv_phase_probe.py).
  const size_t img_elems = (size_t)a.H * a.W * a.y_cs;
  if (a.ksplit > 1 || ((a.Cout & 3) == 0 && (a.y_cs & 3) == 0)) {
    const bool split = a.ksplit > 1;       // split-K: raw partial sums, dense [part][n][pixel][Cout]; epilogue in the finalize kernel
    const int cs = split ? a.Cout : a.y_cs;
    const size_t out_elems = (size_t)a.H * a.W * cs;
    float* const anyp = const_cast<float*>(a.wpk);
    float* obase = split ? a.part + ((size_t)blockIdx.y * a.N + n) * out_elems : a.y + n * out_elems;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(obase, 0, (int)(out_elems * 4), 0x00020000);
    const bool use_mask = !split && a.mask != nullptr;
    const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(use_mask ? a.mask + n * img_elems : anyp), 0,
                                                                         use_mask ? (int)(img_elems * 4) : 0, 0x00020000);
    const bool use_bias = !split && a.bias != nullptr;
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(use_bias ? a.bias : anyp), 0, use_bias ? a.Cout * 4 : 0, 0x00020000);
    const bool relu = !split && a.relu;
#pragma unroll
    for (int ni = 0; ni < C::WN; ++ni) {
      const int cb = co0 + (wn * C::WN + ni) * 32 + 4 * lh;
      f32x4 bv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) bv[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, (cb + 8 * q) * 4, 0, 0));
#pragma unroll
      for (int mi = 0; mi < C::WM; ++mi) {
        const int mb = wm * C::WM + mi;
        const int oy = y0 + (mb / C::TBX) * C::RBH + li / C::RBW;
        const int ox = x0 + (mb % C::TBX) * C::RBW + li % C::RBW;
        const bool inside = oy < a.H && ox < a.W;
        const unsigned pix = inside ? (unsigned)((oy * a.W + ox) * cs) * 4u : OOB;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int co = cb + 8 * q;
          const unsigned off = co < a.Cout ? pix + (unsigned)co * 4u : OOB;
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = acc[mi][ni][4 * q + e] + bv[q][e];
            if (relu) v[e] = v[e] > 0.f ? v[e] : 0.f;
          }
          if (use_mask) {
            const f32x4 m = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(mrs, off, 0, 0));
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = m[e] > 0.f ? v[e] : 0.f;
          }
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, off, 0, 0);
        }
      }
    }
  } else {      // ragged channel counts (the 3-channel input gradient): element-wise
#pragma unroll
    for (int ni = 0; ni < C::WN; ++ni)
#pragma unroll
      for (int mi = 0; mi < C::WM; ++mi) {
        const int mb = wm * C::WM + mi;
        const int oy = y0 + (mb / C::TBX) * C::RBH + li / C::RBW;
        const int ox = x0 + (mb % C::TBX) * C::RBW + li % C::RBW;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co0 + (wn * C::WN + ni) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (co < a.Cout && oy < a.H && ox < a.W) {
            const size_t o = n * img_elems + ((size_t)oy * a.W + ox) * a.y_cs + co;
            float v = acc[mi][ni][r] + (a.bias != nullptr ? a.bias[co] : 0.f);
            if (a.relu) v = v > 0.f ? v : 0.f;
            if (a.mask != nullptr) v = a.mask[o] > 0.f ? v : 0.f;
            a.y[o] = v;
          }
        }
      }
  }
#ifdef OSVOS_CONV_PROF
  PROF_MARK(7);
  if (a.prof != nullptr && lane == 0) {
    unsigned long long* q = a.prof + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 10;
    for (int k = 0; k < 8; ++k) q[k] = pt[k];
    q[8] = t_begin;
    q[9] = tp;
  }
#endif
}

template <class C>
int launch_cfg(const ConvArgs& a0, hipStream_t stream) {
  static bool attr_set_dev[OSVOS_MAX_DEVICES] = {};      // hipFuncSetAttribute is per device
  bool& attr_set = attr_set_dev[osvos_current_device()];
  if (!attr_set) {
    OSVOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_f32_kernel<C>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
    attr_set = true;
  }
  ConvArgs a = a0;
  a.tiles_x = ceil_div(a.W, C::TW);
  a.tiles_y = ceil_div(a.H, C::TH);
  a.nct = ceil_div(a.CoutP, C::BN);
  a.nsp = a.tiles_x * a.tiles_y * a.N;
  const long blocks = a.map == 0 ? (long)a.nct * a.nsp : (long)a.nct * ((a.nsp + 7) / 8) * 8;
  OSVOS_ARG_CHECK(blocks > 0 && blocks < (1L << 31), "conv3x3: grid of %ld blocks", blocks);
  hipLaunchKernelGGL(conv3x3_f32_kernel<C>, dim3((unsigned)blocks, (unsigned)a.ksplit), dim3(256), C::LDS_BYTES, stream, a);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

struct TileInfo {
  int tw, th, bn, wm, wn;
  size_t lds;
};

//                RBW TBX TBY NB WGM WGN
using T0 = Cfg<32, 1, 8, 4, 2, 2>;   // 256 px x 128 co
using T1 = Cfg<32, 1, 8, 2, 4, 1>;   // 256 px x  64 co
using T2 = Cfg<32, 1, 4, 2, 2, 2>;   // 128 px x  64 co
using T3 = Cfg<32, 1, 8, 1, 4, 1>;   // 256 px x  32 co
using T4 = Cfg<16, 1, 4, 2, 2, 2>;   // 16x8 px x 64 co
using T5 = Cfg<8, 1, 2, 2, 2, 2>;    //  8x8 px x 64 co
using T6 = Cfg<16, 1, 2, 2, 2, 2>;   // 16x4 px x 64 co
using T7 = Cfg<32, 1, 2, 2, 2, 2>;   // 32x2 px x 64 co
using T8 = Cfg<32, 1, 4, 4, 2, 2>;   // 128 px x 128 co
using T9 = Cfg<16, 1, 4, 1, 4, 1>;   // 16x8 px x 32 co
// shapes for the narrow deep feature maps (107 and 54 pixels wide at 480p): 32-wide tiles waste 16-20 % there
using T10 = Cfg<8, 2, 4, 1, 4, 1>;   // 16x16 px x 32 co
using T11 = Cfg<8, 1, 4, 1, 4, 1>;   //  8x16 px x 32 co
using T12 = Cfg<16, 1, 8, 1, 4, 1>;  // 16x16 px x 32 co (16x2 row blocks)
using T13 = Cfg<8, 2, 3, 2, 2, 2>;   // 16x12 px x 64 co
using T14 = Cfg<32, 1, 4, 1, 4, 1>;  // 32x4 px x 32 co: T9's shape with 32-wide row blocks (conflict-free LDS reads)
constexpr int kNumTiles = 15;

template <class C>
constexpr TileInfo info() { return TileInfo{C::TW, C::TH, C::BN, C::WM, C::WN, C::LDS_BYTES}; }
const TileInfo kTiles[kNumTiles] = {info<T0>(), info<T1>(), info<T2>(), info<T3>(), info<T4>(),
                                    info<T5>(), info<T6>(), info<T7>(), info<T8>(), info<T9>(),
                                    info<T10>(), info<T11>(), info<T12>(), info<T13>(), info<T14>()};

// Measured on MI355X (tools/tune_conv.py, profiles/r01_tune_conv_tiles.txt): the 128-pixel x 32-cout tile T9 -- one
// accumulator per wave, 5-7 workgroups per CU -- is the fastest or within 2 % of the fastest on every layer of the
// network, forward and data-gradient, once the epilogue is a handful of 16-byte stores.  The big register-blocked
// tiles (T0/T1/T8) run one or two waves per SIMD and expose their prologue, epilogue and barriers.
int pick_tile(int N, int H, int W, int Cin, int CoutP) {
  (void)N; (void)H; (void)W; (void)Cin; (void)CoutP;
  return 9;
}

// y = epi(bias + sum_k part[k]) for the split-K launches: Cout % 4 == 0, y channel stride y_cs
__global__ void conv_splitk_finalize_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                            const float* __restrict__ mask, float* __restrict__ y, long npix, int Cout,
                                            int y_cs, int ksplit, int relu) {
  const long total4 = npix * (Cout / 4);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const long pix = i / (Cout / 4);
    const int c4 = (int)(i % (Cout / 4)) * 4;
    f32x4 s = *reinterpret_cast<const f32x4*>(part + pix * Cout + c4);
    for (int k = 1; k < ksplit; ++k) s += *reinterpret_cast<const f32x4*>(part + ((size_t)k * npix + pix) * Cout + c4);
    if (bias != nullptr) s += *reinterpret_cast<const f32x4*>(bias + c4);
    const size_t o = (size_t)pix * y_cs + c4;
    if (relu)
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] = s[e] > 0.f ? s[e] : 0.f;
    if (mask != nullptr) {
      const f32x4 m = *reinterpret_cast<const f32x4*>(mask + o);
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] = m[e] > 0.f ? s[e] : 0.f;
    }
    *reinterpret_cast<f32x4*>(y + o) = s;
  }
}

// how many K parts a launch should be cut into so the machine sees >= ~7 workgroups per CU (balance) without
// drowning in partial-sum traffic; 1 for the big shallow layers
int pick_ksplit(const TileInfo& t, int N, int H, int W, int Cin, int Cout, int CoutP, int y_cs) {
  if (Cout % 4 != 0 || y_cs % 4 != 0 || Cin < 256) return 1;
  const long blocks = (long)N * ceil_div(H, t.th) * ceil_div(W, t.tw) * ceil_div(CoutP, t.bn);
  int ks = 1;
  // measured (tools/tune_splitk.py, end-of-round kernels): grids of about one workgroup per CU gain (conv5_x at batch 1, 0.090 -> 0.088
  // ms).  conv4_x (896 workgroups, 3.5 per CU) would gain too (0.288 -> 0.269 ms, +1.7 % on the step) but is left un-split on purpose:
  // the changed summation order moves the stage-0 gradients of the un-trained test net by ~1e-3 (ReLU / arg-max flips at near-ties),
  // past the parity bar of tests/test_gpu_net.py::test_full_size_against_cpu_oracle.  OSVOS_CONV_KSPLIT=2 forces it.
  while (ks < 8 && blocks * ks < 512 && (Cin / 8) / (ks * 2) >= 8) ks *= 2;
  // skinny Cout (side_prep, CoutP = 32): a few dozen workgroups at most -- cut K as far as 4-chunk parts allow
  if (CoutP <= 32) while (ks < 8 && blocks * ks < 256 && (Cin / 8) / (ks * 2) >= 2) ks *= 2;
  return ks;
}

}  // namespace

extern "C" int osvos_conv3x3_num_tiles(void) { return kNumTiles; }

size_t osvos_conv3x3_splitk_ws_bytes_f32(int N, int H, int W, int Cout) {
  return align_up((size_t)8 * N * H * W * Cout * sizeof(float), 256);      // up to 8 K parts
}

// part_ws: NULL (never split) or a buffer of osvos_conv3x3_splitk_ws_bytes_f32() for the split-K partial sums
// phase-counter hook of tools/conv_phase_probe.py: exists only in probe builds (make EXTRA=-DOSVOS_CONV_PROF); the shipped library has no
// process-global device pointer behind its re-entrant ABI
#ifdef OSVOS_CONV_PROF
static unsigned long long* g_conv_prof_f32 = nullptr;
extern "C" void osvos_debug_set_conv_prof_f32(void* p) { g_conv_prof_f32 = (unsigned long long*)p; }
#define OSVOS_CONV_PROF_PTR g_conv_prof_f32
#else
#define OSVOS_CONV_PROF_PTR nullptr
#endif
static thread_local int g_force_ksplit = 0;      // tests / tuning: osvos_conv3x3_splitk(..., ksplit > 0, ...)
void osvos_conv3x3_force_ksplit(int k) { g_force_ksplit = k; }

int osvos_conv3x3_f32(const float* x, const float* wpk, const float* bias, const float* mask, float* y,
                      int N, int H, int W, int Cin, int Cout, int y_cs, int relu, int tile, hipStream_t stream) {
  return osvos_conv3x3_f32_ws(x, wpk, bias, mask, y, N, H, W, Cin, Cout, y_cs, relu, tile, nullptr, stream);
}

int osvos_conv3x3_f32_ws(const float* x, const float* wpk, const float* bias, const float* mask, float* y,
                         int N, int H, int W, int Cin, int Cout, int y_cs, int relu, int tile, void* part_ws, hipStream_t stream) {
  OSVOS_ARG_CHECK(x && wpk && y, "conv3x3: null pointer");
  OSVOS_ARG_CHECK(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv3x3: bad shape");
  OSVOS_ARG_CHECK(Cin % 8 == 0, "conv3x3 f32: Cin (%d) must be a multiple of 8 (pad the input)", Cin);
  OSVOS_ARG_CHECK(y_cs >= Cout, "conv3x3: y channel stride %d < Cout %d", y_cs, Cout);
  OSVOS_ARG_CHECK((long)H * W * Cin < (1L << 29) && (long)H * W * y_cs < (1L << 29), "conv3x3: image too large for 31-bit byte offsets");
  // f32x3: the same fp32 problem on the bf16 matrix pipe with three-way split operands (conv3x3_f32x3.hip).  Tile ids
  // 200.. force it (tests, tuning); tile -2 = "automatic, in the f32x3 arithmetic where it applies" (what the callers that were handed
  // dtype OSVOS_F32_X3 pass; round 4: this replaces a process-wide mutable mode -- the arithmetic is a per-call argument of the ABI).
  if ((tile >= 200 || (tile == -2 && osvos_conv3x3_f32x3_applicable(Cin, Cout, y_cs))))
    return osvos_conv3x3_f32x3(x, wpk, bias, mask, y, N, H, W, Cin, Cout, y_cs, relu, tile >= 200 ? tile - 200 : -1, g_force_ksplit, part_ws, stream);
  ConvArgs a;
  a.x = x; a.wpk = wpk; a.bias = bias; a.mask = mask; a.y = y;
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.CoutP = osvos_cout_pad(Cout); a.y_cs = y_cs;
  a.relu = relu;
  if (tile < 0) {
    OSVOS_ENV_INT(env_tile, "OSVOS_CONV_TILE", -1);
    tile = env_tile >= 0 ? env_tile : pick_tile(N, H, W, Cin, a.CoutP);
    // activations larger than the weights -> keep the halo tile XCD-local
    if (env_tile < 0 && (double)H * W * Cin > 9.0 * Cin * a.CoutP) tile += 100;
  }
  a.map = tile >= 100 ? 1 : 0;
  tile %= 100;
  OSVOS_ARG_CHECK(tile >= 0 && tile < kNumTiles, "conv3x3: unknown tile config %d", tile);
  a.ksplit = 1;
  a.prof = OSVOS_CONV_PROF_PTR;
  a.part = reinterpret_cast<float*>(part_ws);
  if (part_ws != nullptr) {
    // OSVOS_CONV_KSPLIT overrides the automatic choice, but only where the automatic choice could split as well (Cin >= 256):
    // the caller sizes `part_ws` for those layers only (net.cpp ws_layout), a forced split of a shallow layer would overrun it
    OSVOS_ENV_INT(env_ks, "OSVOS_CONV_KSPLIT", 0);
    a.ksplit = g_force_ksplit > 0 ? g_force_ksplit
                                  : ((env_ks > 0 && Cin >= 256) ? env_ks : pick_ksplit(kTiles[tile], N, H, W, Cin, Cout, a.CoutP, y_cs));
    if (a.ksplit < 1 || a.ksplit > 8 || Cout % 4 != 0 || y_cs % 4 != 0 || a.ksplit > (Cin >> 3)) a.ksplit = 1;
  }
  int rc;
  switch (tile) {
    case 0: rc = launch_cfg<T0>(a, stream); break;
    case 1: rc = launch_cfg<T1>(a, stream); break;
    case 2: rc = launch_cfg<T2>(a, stream); break;
    case 3: rc = launch_cfg<T3>(a, stream); break;
    case 4: rc = launch_cfg<T4>(a, stream); break;
    case 5: rc = launch_cfg<T5>(a, stream); break;
    case 6: rc = launch_cfg<T6>(a, stream); break;
    case 7: rc = launch_cfg<T7>(a, stream); break;
    case 8: rc = launch_cfg<T8>(a, stream); break;
    case 9: rc = launch_cfg<T9>(a, stream); break;
    case 10: rc = launch_cfg<T10>(a, stream); break;
    case 11: rc = launch_cfg<T11>(a, stream); break;
    case 12: rc = launch_cfg<T12>(a, stream); break;
    case 13: rc = launch_cfg<T13>(a, stream); break;
    case 14: rc = launch_cfg<T14>(a, stream); break;
    default: osvos_set_error("conv3x3: unknown tile config %d", tile); return -1;
  }
  if (rc || a.ksplit == 1) return rc;
  return osvos_conv3x3_splitk_finalize_f32(a.part, bias, mask, y, (long)N * H * W, Cout, y_cs, a.ksplit, relu, stream);
}

int osvos_conv3x3_splitk_finalize_f32(const float* part, const float* bias, const float* mask, float* y, long npix, int Cout, int y_cs,
                                      int ksplit, int relu, hipStream_t stream) {
  long blocks = (npix * (Cout / 4) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(conv_splitk_finalize_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, part, bias, mask, y, npix, Cout, y_cs, ksplit, relu);
  OSVOS_LAUNCH_CHECK();
  return 0;
}


#ifdef OSVOS_CONV_PROF   // C entry points of the scratch library tools/conv_phase_probe.py builds from this file alone
__global__ void prof_pack_fwd_f32_kernel(const float* __restrict__ w, float* __restrict__ wpk, int Cout, int Cin, int CinP, int CoutP) {
  const long total = 9L * CinP * CoutP;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 3);
    long t = i >> 2;
    const int co = (int)(t % CoutP);
    t /= CoutP;
    const int cq = (int)(t % (CinP / 4));
    const int tap = (int)(t / (CinP / 4));
    const int ci = cq * 4 + e;
    wpk[i] = (co < Cout && ci < Cin) ? w[((long)co * Cin + ci) * 9 + tap] : 0.f;
  }
}
extern "C" int osvos_prof_pack_fwd_f32(const float* w, float* wpk, int Cout, int Cin) {
  hipLaunchKernelGGL(prof_pack_fwd_f32_kernel, dim3(1024), dim3(256), 0, nullptr, w, wpk, Cout, Cin, (Cin + 7) / 8 * 8, osvos_cout_pad(Cout));
  return (int)hipGetLastError();
}
extern "C" int osvos_prof_conv3x3_f32(const float* x, const float* wpk, float* y, int N, int H, int W, int Cin, int Cout, int tile) {
  return osvos_conv3x3_f32(x, wpk, nullptr, nullptr, y, N, H, W, Cin, Cout, Cout, 1, tile, nullptr);
}
#endif
