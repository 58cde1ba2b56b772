// 3x3 convolution with bf16 MFMA operands: fp32 NHWC tensors in HBM, operands rounded to bf16
// (round-to-nearest-even, v_cvt_pk_bf16_f32) while they are staged into LDS, fp32 accumulation in
// v_mfma_f32_32x32x16_bf16, fp32 result.  ("dtype" OSVOS_F32_BF16MFMA.)  Same implicit-GEMM
// structure as conv3x3_f32.hip -- halo tile staged once per K chunk and re-used by the 9 taps --
// with the quantities rescaled for a matrix pipe that is 16x faster:
//   * one 16-byte LDS group = 8 bf16 channels; a K chunk = 4 groups = 32 channels = 2 MFMA k-steps,
//     lane l reads groups 2*ks + (l>>5): ONE ds_read_b128 per operand per MFMA
//   * big register-blocked wave tiles (up to 4x2 accumulators): at 2.5 PFLOP/s the LDS read rate
//     and the L2->LDS weight stream, not the matrix pipe, are what has to be rationed
//   * numerics equal a bf16-activation pipeline: rounding happens right before the MFMA either way
#include "common.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

struct ConvArgsB {
  const float* x;        // fp32 NHWC, channel stride Cin (multiple of 8)
  const uint4* wpk;      // bf16 pack [9][CinP/8][CoutP][8], CinP = Cin rounded up to 32
  const float* bias;
  const float* mask;
  float* y;
  int N, H, W, Cin, CinP, Cout, CoutP, y_cs;
  int tiles_x, tiles_y, nct, nsp, map;
  int relu;
};

constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
constexpr int pitch_for(int rbw, int hw) {
  return rbw == 32 ? hw : (rbw == 16 ? cdiv(hw, 16) * 16 : cdiv(hw - 8, 16) * 16 + 8);
}

constexpr int KG = 4;   // 16-byte groups (8 channels each) per K chunk

template <int RBW_, int TBX_, int TBY_, int NB_, int WGM_, int WGN_, int PIPE_ = 1>
struct CfgB {
  static constexpr int RBW = RBW_, TBX = TBX_, TBY = TBY_, NB = NB_, WGM = WGM_, WGN = WGN_, PIPE = PIPE_;
  static constexpr int RBH = 32 / RBW;
  static constexpr int TW = TBX * RBW, TH = TBY * RBH;
  static constexpr int HWD = TW + 2, HHT = TH + 2;
  static constexpr int PITCH = pitch_for(RBW, HWD);
  static constexpr int PLANE = HHT * PITCH;
  static constexpr int BN = NB * 32;
  static constexpr int A_U4 = KG * PLANE;              // uint4 (16 B) slots
  static constexpr int B_U4 = 9 * KG * BN;
  static constexpr int BUF_U4 = A_U4 + B_U4;
  static constexpr int A_LOAD = HHT * HWD * KG;
  static constexpr int NT = 64 * WGM * WGN;            // 4 or 8 waves per workgroup
  static constexpr int NA = cdiv(A_LOAD, NT);
  static constexpr int NBL = cdiv(B_U4, NT);
  static constexpr int MB = TBX * TBY;
  static constexpr int WM = MB / WGM, WN = NB / WGN;
  static constexpr size_t LDS_BYTES = (size_t)BUF_U4 * 16;   // single buffer; the next chunk waits in registers
  static_assert(WGM * WGN == 4 || WGM * WGN == 8, "4 or 8 waves per workgroup");
  static_assert(MB % WGM == 0 && NB % WGN == 0, "wave grid must divide the tile");
};

__device__ inline uint4 pack_bf16x8(const f32x4& a, const f32x4& b) {
  bf16x8_t v;
  v[0] = (__bf16)a[0]; v[1] = (__bf16)a[1]; v[2] = (__bf16)a[2]; v[3] = (__bf16)a[3];
  v[4] = (__bf16)b[0]; v[5] = (__bf16)b[1]; v[6] = (__bf16)b[2]; v[7] = (__bf16)b[3];
  return __builtin_bit_cast(uint4, v);
}

template <class C>
__global__ __launch_bounds__(C::NT) void conv3x3_bf16_kernel(ConvArgsB a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* lds = reinterpret_cast<uint4*>(smem);
  uint4* As = lds;
  uint4* Bs = lds + C::A_U4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave / C::WGN, wn = wave % C::WGN;

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
  const int CG = a.CinP >> 3;                     // 8-channel groups in the weight pack
  const float* ximg = a.x + (size_t)n * a.H * a.W * a.Cin;

  int a_src[C::NA], a_dst[C::NA];                // element offset of group 0, -1 zero fill, -2 none
  int a_grp[C::NA];
#pragma unroll
  for (int i = 0; i < C::NA; ++i) {
    const int e = tid + i * C::NT;
    a_src[i] = -2; a_dst[i] = 0; a_grp[i] = 0;
    if (e < C::A_LOAD) {
      const int g = e % KG, pix = e / KG;
      const int hy = pix / C::HWD, hx = pix % C::HWD;
      const int gy = y0 + hy - 1, gx = x0 + hx - 1;
      a_dst[i] = g * C::PLANE + hy * C::PITCH + hx;
      a_grp[i] = g;
      a_src[i] = (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? ((gy * a.W + gx) * a.Cin + 8 * g) : -1;
    }
  }
  int b_src[C::NBL];
#pragma unroll
  for (int i = 0; i < C::NBL; ++i) {
    const int e = tid + i * C::NT;
    b_src[i] = -2;
    if (e < C::B_U4) {
      const int tap = e / (KG * C::BN), rem = e % (KG * C::BN);
      const int g = rem / C::BN, nn = rem % C::BN;
      b_src[i] = (co0 + nn < a.CoutP) ? (tap * CG + g) * a.CoutP + co0 + nn : -1;
    }
  }

  f32x4 ra[C::NA][2];
  uint4 rb[C::NBL];
  auto load_chunk = [&](int kc) {
    const int c0 = kc * 8 * KG;
#pragma unroll
    for (int i = 0; i < C::NA; ++i) {
      f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
      if (a_src[i] >= 0 && c0 + 8 * a_grp[i] < a.Cin) {
        const float* p = ximg + a_src[i] + c0;
        v0 = *reinterpret_cast<const f32x4*>(p);
        v1 = *reinterpret_cast<const f32x4*>(p + 4);
      }
      ra[i][0] = v0;
      ra[i][1] = v1;
    }
    const uint4* wq = a.wpk + (size_t)(c0 >> 3) * a.CoutP;
#pragma unroll
    for (int i = 0; i < C::NBL; ++i) {
      uint4 v = {0u, 0u, 0u, 0u};
      if (b_src[i] >= 0) v = wq[b_src[i]];
      rb[i] = v;
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < C::NA; ++i)
      if (a_src[i] != -2) As[a_dst[i]] = pack_bf16x8(ra[i][0], ra[i][1]);
#pragma unroll
    for (int i = 0; i < C::NBL; ++i)
      if (b_src[i] != -2) Bs[tid + i * C::NT] = rb[i];
  };

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

  const int nchunks = a.CinP / (8 * KG);
  load_chunk(0);
  for (int kc = 0; kc < nchunks; ++kc) {
    __syncthreads();                     // every wave is done with the previous chunk's tiles
    store_chunk();
    __syncthreads();
    if (kc + 1 < nchunks) load_chunk(kc + 1);      // in flight during the MFMAs below
    // 18 (tap, k-step) stages, software pipelined: operands of stage s+1 are requested before the
    // MFMAs of stage s issue; sched_barrier pins that order
    uint4 fa[1 + C::PIPE][C::WM], fb[1 + C::PIPE][C::WN];
    auto ldfrag = [&](int st, int set) {
      const int tap = st >> 1, ks = st & 1;
      const int r = tap / 3, s = tap % 3;
#pragma unroll
      for (int mi = 0; mi < C::WM; ++mi) fa[set][mi] = As[a_idx[mi] + 2 * ks * C::PLANE + r * C::PITCH + s];
#pragma unroll
      for (int ni = 0; ni < C::WN; ++ni) fb[set][ni] = Bs[b_idx + (tap * KG + 2 * ks) * C::BN + ni * 32];
    };
    if (C::PIPE) ldfrag(0, 0);
#pragma unroll
    for (int st = 0; st < 18; ++st) {
      const int cur = C::PIPE ? (st & 1) : 0;
      if (C::PIPE) {
        if (st + 1 < 18) ldfrag(st + 1, (st + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
      } else {
        ldfrag(st, 0);     // 8-wave tiles: two waves per SIMD cover each other's LDS latency, registers are the scarce resource
      }
#pragma unroll
      for (int mi = 0; mi < C::WM; ++mi)
#pragma unroll
        for (int ni = 0; ni < C::WN; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[cur][mi]),
                                                                __builtin_bit_cast(bf16x8_t, fb[cur][ni]), acc[mi][ni], 0, 0, 0);
      if (C::PIPE) __builtin_amdgcn_sched_barrier(0);
    }
  }

#pragma unroll
  for (int ni = 0; ni < C::WN; ++ni) {
    const int co = co0 + (wn * C::WN + ni) * 32 + li;
    const bool co_ok = co < a.Cout;
    const float bv = (a.bias != nullptr && co_ok) ? a.bias[co] : 0.f;
#pragma unroll
    for (int mi = 0; mi < C::WM; ++mi) {
      const int mb = wm * C::WM + mi;
      const int mbx = mb % C::TBX, mby = mb / C::TBX;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int prow = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int oy = y0 + mby * C::RBH + prow / C::RBW;
        const int ox = x0 + mbx * C::RBW + prow % C::RBW;
        if (co_ok && oy < a.H && ox < a.W) {
          const size_t o = ((size_t)(n * a.H + oy) * a.W + ox) * a.y_cs + co;
          float v = acc[mi][ni][r] + bv;
          if (a.relu) v = v > 0.f ? v : 0.f;
          if (a.mask != nullptr) v = a.mask[o] > 0.f ? v : 0.f;
          a.y[o] = v;
        }
      }
    }
  }
}

template <class C>
int launch_cfg(const ConvArgsB& a0, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    OSVOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_bf16_kernel<C>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
    attr_set = true;
  }
  ConvArgsB a = a0;
  a.tiles_x = ceil_div(a.W, C::TW);
  a.tiles_y = ceil_div(a.H, C::TH);
  a.nct = ceil_div(a.CoutP, C::BN);
  a.nsp = a.tiles_x * a.tiles_y * a.N;
  const long blocks = a.map == 0 ? (long)a.nct * a.nsp : (long)a.nct * ((a.nsp + 7) / 8) * 8;
  OSVOS_ARG_CHECK(blocks > 0 && blocks < (1L << 31), "conv3x3 bf16: grid of %ld blocks", blocks);
  hipLaunchKernelGGL(conv3x3_bf16_kernel<C>, dim3((unsigned)blocks), dim3(C::NT), C::LDS_BYTES, stream, a);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

struct TileInfoB { int tw, th, bn, wm, wn; size_t lds; };

//                 RBW TBX TBY NB WGM WGN
using B0 = CfgB<32, 1, 8, 4, 2, 2>;   // 256 px x 128 co, 4x2 accumulators per wave
using B1 = CfgB<32, 1, 8, 2, 2, 2>;   // 256 px x  64 co, 4x1
using B2 = CfgB<32, 1, 4, 4, 2, 2>;   // 128 px x 128 co, 2x2
using B3 = CfgB<32, 1, 4, 2, 2, 2>;   // 128 px x  64 co, 2x1
using B4 = CfgB<16, 1, 4, 4, 2, 2>;   // 16x8 px x 128 co, 2x2
using B5 = CfgB<16, 1, 4, 2, 2, 2>;   // 16x8 px x  64 co, 2x1
using B6 = CfgB<32, 1, 8, 1, 4, 1>;   // 256 px x  32 co, 2x1
using B7 = CfgB<8, 1, 2, 2, 2, 2>;    //  8x8 px x  64 co, 1x1
// (a 512 px x 128 co, 4x2-accumulator 8-wave tile would need > 256 registers per lane: it spills)
using B8 = CfgB<32, 1, 16, 2, 4, 2>;  // 512 px x  64 co, 8 waves, 4x1
using B9 = CfgB<32, 1, 8, 4, 4, 2>;   // 256 px x 128 co, 8 waves, 2x2
constexpr int kNumTilesB = 10;
template <class C>
constexpr TileInfoB infoB() { return TileInfoB{C::TW, C::TH, C::BN, C::WM, C::WN, C::LDS_BYTES}; }
const TileInfoB kTilesB[kNumTilesB] = {infoB<B0>(), infoB<B1>(), infoB<B2>(), infoB<B3>(), infoB<B4>(), infoB<B5>(), infoB<B6>(), infoB<B7>(),
                                       infoB<B8>(), infoB<B9>()};

// Measured (profiles/r01_tune_bf16_*.txt): B1 (256 px x 64 couts, 4 accumulators, 2 workgroups per CU) wins
// whenever it yields enough workgroups (up to 825 TFLOP/s on conv3_x/conv4_x at batch 12); the 8-accumulator
// B0 tile runs one wave per SIMD and loses by 30 %; small frames fall back to 128- and 64-pixel tiles.
int pick_tile_b(int N, int H, int W, int CoutP) {
  if (CoutP <= 32) return 6;
  const int order[] = {1, 5, 7};
  for (int k = 0; k < 3; ++k) {
    const TileInfoB& t = kTilesB[order[k]];
    const long tiles = (long)N * ceil_div(H, t.th) * ceil_div(W, t.tw) * ceil_div(CoutP, t.bn);
    if (tiles >= 400 || k == 2) return order[k];
  }
  return 7;
}

// wpk[((tap*CG + cg)*CoutP + co)*8 + e] = bf16(W[co][8cg+e][tap])   (zero padded)
__global__ void pack_fwd_bf16_kernel(const float* __restrict__ w, bf16_t* __restrict__ wpk, int Cout, int Cin, int CinP, int CoutP) {
  const int CG = CinP / 8;
  const long total = 9L * CG * CoutP * 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7);
    long t = i >> 3;
    const int co = (int)(t % CoutP);
    t /= CoutP;
    const int cg = (int)(t % CG);
    const int tap = (int)(t / CG);
    const int ci = cg * 8 + e;
    wpk[i] = (co < Cout && ci < Cin) ? f32_to_bf16(w[((long)co * Cin + ci) * 9 + tap]) : (bf16_t)0;
  }
}

// data-gradient pack: roles swapped, filter rotated by 180 degrees
__global__ void pack_dgrad_bf16_kernel(const float* __restrict__ w, bf16_t* __restrict__ wpk, int Cout, int Cin, int CoutK, int CinP) {
  const int CG = CoutK / 8;
  const long total = 9L * CG * CinP * 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7);
    long t = i >> 3;
    const int ci = (int)(t % CinP);
    t /= CinP;
    const int cog = (int)(t % CG);
    const int tap = (int)(t / CG);
    const int co = cog * 8 + e;
    wpk[i] = (co < Cout && ci < Cin) ? f32_to_bf16(w[((long)co * Cin + ci) * 9 + (8 - tap)]) : (bf16_t)0;
  }
}

inline int grid_for(long total) {
  long b = (total + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

int osvos_conv3x3_bf16mfma_num_tiles(void) { return kNumTilesB; }

int osvos_pack_fwd_bf16(const float* w, void* wpk, int Cout, int Cin, hipStream_t stream) {
  OSVOS_ARG_CHECK(w && wpk && Cout > 0 && Cin > 0, "pack_fwd bf16: bad arguments");
  const int CinP = (Cin + 31) / 32 * 32, CoutP = osvos_cout_pad(Cout);
  hipLaunchKernelGGL(pack_fwd_bf16_kernel, dim3(grid_for(9L * CinP * CoutP)), dim3(256), 0, stream, w, (bf16_t*)wpk, Cout, Cin, CinP, CoutP);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

int osvos_pack_dgrad_bf16(const float* w, void* wpk, int Cout, int Cin, hipStream_t stream) {
  OSVOS_ARG_CHECK(w && wpk && Cout > 0 && Cin > 0, "pack_dgrad bf16: bad arguments");
  const int CoutK = (Cout + 31) / 32 * 32, CinP = osvos_cout_pad(Cin);
  hipLaunchKernelGGL(pack_dgrad_bf16_kernel, dim3(grid_for(9L * CoutK * CinP)), dim3(256), 0, stream, w, (bf16_t*)wpk, Cout, Cin, CoutK, CinP);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

// x fp32 NHWC (stride Cin, multiple of 8), wpk from osvos_pack_{fwd,dgrad}_bf16 with the same Cin/Cout roles
int osvos_conv3x3_bf16mfma(const float* x, const void* wpk, const float* bias, const float* mask, float* y,
                           int N, int H, int W, int Cin, int Cout, int y_cs, int relu, int tile, hipStream_t stream) {
  OSVOS_ARG_CHECK(x && wpk && y, "conv3x3 bf16: null pointer");
  OSVOS_ARG_CHECK(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv3x3 bf16: bad shape");
  OSVOS_ARG_CHECK(Cin % 8 == 0, "conv3x3 bf16: Cin (%d) must be a multiple of 8", Cin);
  OSVOS_ARG_CHECK(y_cs >= Cout, "conv3x3 bf16: y channel stride %d < Cout %d", y_cs, Cout);
  OSVOS_ARG_CHECK((long)H * W * Cin < (1L << 31), "conv3x3 bf16: image too large for 32-bit offsets");
  ConvArgsB a;
  a.x = x; a.wpk = reinterpret_cast<const uint4*>(wpk); a.bias = bias; a.mask = mask; a.y = y;
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.CinP = (Cin + 31) / 32 * 32; a.Cout = Cout; a.CoutP = osvos_cout_pad(Cout); a.y_cs = y_cs;
  a.relu = relu;
  if (tile < 0) {
    const char* env = getenv("OSVOS_CONV_TILE_BF16");
    tile = env ? atoi(env) : pick_tile_b(N, H, W, a.CoutP);
    if (!env && (double)H * W * Cin * 4 > 9.0 * Cin * a.CoutP * 2) tile += 100;
  }
  a.map = tile >= 100 ? 1 : 0;
  tile %= 100;
  switch (tile) {
    case 0: return launch_cfg<B0>(a, stream);
    case 1: return launch_cfg<B1>(a, stream);
    case 2: return launch_cfg<B2>(a, stream);
    case 3: return launch_cfg<B3>(a, stream);
    case 4: return launch_cfg<B4>(a, stream);
    case 5: return launch_cfg<B5>(a, stream);
    case 6: return launch_cfg<B6>(a, stream);
    case 7: return launch_cfg<B7>(a, stream);
    case 8: return launch_cfg<B8>(a, stream);
    case 9: return launch_cfg<B9>(a, stream);
    default: osvos_set_error("conv3x3 bf16: unknown tile config %d", tile); return -1;
  }
}
