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
#include "maskbits.h"
#include <type_traits>
#include "kernels.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct ConvArgsB {
  const void* x;         // NHWC, channel stride Cin (multiple of 8): fp32 (XB = 0) or bf16 (XB = 1)
  const uint4* wpk;      // bf16 pack [9][CinP/8][CoutP][8], CinP = Cin rounded up to 32
  const float* bias;
  const void* mask;      // NHWC like y: fp32, or bf16 when mask_bf16 != 0
  float* y;              // may be NULL when ybf is given (bf16-only result)
  bf16_t* ybf;           // optional bf16 copy of y (same channel stride): the next convolution's operand
  int N, H, W, Cin, CinP, Cout, CoutP, y_cs;
  int tiles_x, tiles_y, nct, nsp, map;
  int relu, mask_bf16;
  const unsigned* mask_bits;   // ReLU mask as ONE BIT per element ([N][H][W][y_cs / 32] words, bit = channel % 32; maskbits.h): takes precedence over `mask`
  unsigned* y_bits;            // optional: the same for the result (written next to y / ybf by the forward of a layer whose output is a later mask)
  bf16_t* pooled;              // optional: maxpool2x2 (ceil mode) of the bf16 result, [N][ceil(H/2)][ceil(W/2)][y_cs] (last convolution of a stage; needs ReLU)
  unsigned char* pool_code;    // optional, with pooled: one byte per pooled element for the pool's backward (pool.hip: first-max position + 4 "input > 0" bits)
  unsigned long long* prof;   // phase cycle counters (only read by builds with -DOSVOS_CONV_PROF; tools/conv_phase_probe.py)
};

constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
constexpr int pitch_for(int rbw, int hw) {
  return rbw == 32 ? hw : (rbw == 16 ? cdiv(hw, 16) * 16 : cdiv(hw - 8, 16) * 16 + 8);
}


template <int RBW_, int TBX_, int TBY_, int NB_, int WGM_, int WGN_, int PIPE_ = 1, int OCC_ = 1, int KG_ = 4>
struct CfgB {
  static constexpr int KG = KG_;                       // 16-byte groups (8 bf16 channels each) per K chunk: 4 = 32 channels, 2 = 16
  static constexpr int KS = KG_ / 2;                   // MFMA k-steps (16 channels) per tap and chunk
  static_assert(KG_ == 2 || KG_ == 4, "chunk of 16 or 32 channels");
  static constexpr int OCC = OCC_;                     // waves per SIMD the register allocation must allow
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
  static constexpr size_t LDS_BYTES = (size_t)(BUF_U4 + 1) * 16;   // single buffer (+ one spare slot); the next chunk waits in registers
  static_assert(WGM * WGN == 4 || WGM * WGN == 8, "4 or 8 waves per workgroup");
  static_assert(MB % WGM == 0 && NB % WGN == 0, "wave grid must divide the tile");
};

__device__ inline uint4 pack_bf16x8(const f32x4& a, const f32x4& b) {
  bf16x8_t v;
  v[0] = (__bf16)a[0]; v[1] = (__bf16)a[1]; v[2] = (__bf16)a[2]; v[3] = (__bf16)a[3];
  v[4] = (__bf16)b[0]; v[5] = (__bf16)b[1]; v[6] = (__bf16)b[2]; v[7] = (__bf16)b[3];
  return __builtin_bit_cast(uint4, v);
}

template <class C, int XB>      // XB = 1: the input tensor is already bf16 (half the bytes, no conversion while staging)
__global__ __launch_bounds__(C::NT, C::OCC) void conv3x3_bf16_kernel(ConvArgsB a) {
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
  constexpr int XE = XB ? 2 : 4;                  // bytes per input element
  const char* ximg = reinterpret_cast<const char*>(a.x) + (size_t)n * a.H * a.W * a.Cin * XE;

  // Staging goes through raw buffer loads: a lane whose halo pixel lies outside the image (or that has no
  // slot at all) gets byte offset 0x80000000 -- beyond num_records -- and the hardware returns zeros.  No
  // per-load branch or select is left in the K loop (hipcc turns every `if (ok) v = *p` into an exec-mask
  // branch pair: 36 of them per chunk before).  The chunk advance rides in the scalar offset.
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ximg), 0, (int)((size_t)a.H * a.W * a.Cin * XE), 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(a.wpk), 0, (int)((size_t)9 * CG * a.CoutP * 16), 0x00020000);
  // (A measured alternative -- consecutive lanes on consecutive 16-byte pieces, 8-byte LDS stores -- was no faster:
  // the cost of the activation loads is their L2 miss rate, not their lane pattern.)
  unsigned a_off[C::NA];                         // byte offset of the slot's 8 channels in chunk 0
  int a_dst[C::NA];
  static_assert(C::NT % C::KG == 0, "a thread keeps the same channel group in every round");
  const int a_grp = tid % C::KG;
#pragma unroll
  for (int i = 0; i < C::NA; ++i) {
    const int e = tid + i * C::NT;
    const int g = e % C::KG, pix = e / C::KG;
    const int hy = pix / C::HWD, hx = pix % C::HWD;
    const int gy = y0 + hy - 1, gx = x0 + hx - 1;
    const bool slot = e < C::A_LOAD;
    a_dst[i] = slot ? g * C::PLANE + hy * C::PITCH + hx : C::BUF_U4;      // lanes without a slot write the spare slot
    a_off[i] = (slot && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? (unsigned)(((gy * a.W + gx) * a.Cin + 8 * g) * XE) : OOB;
  }
  unsigned b_off[C::NBL];
#pragma unroll
  for (int i = 0; i < C::NBL; ++i) {
    const int e = tid + i * C::NT;
    const int tap = e / (C::KG * C::BN), rem = e % (C::KG * C::BN);
    const int g = rem / C::BN, nn = rem % C::BN;
    b_off[i] = (e < C::B_U4 && co0 + nn < a.CoutP) ? (unsigned)(((tap * CG + g) * a.CoutP + co0 + nn) * 16) : OOB;
  }
  const int cin_groups = a.Cin >> 3;             // 8-channel groups that exist in x (the pack is zero padded to CinP)

  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 ra[C::NA][2];
  u32x4 rb[C::NBL];
#ifdef OSVOS_PROF_NO_A       // probe builds only: out-of-range offsets = zero fill without memory traffic
#pragma unroll
  for (int i = 0; i < C::NA; ++i) a_off[i] = OOB;
#endif
#ifdef OSVOS_PROF_NO_B
#pragma unroll
  for (int i = 0; i < C::NBL; ++i) b_off[i] = OOB;
#endif
  auto load_chunk = [&](int kc) {
#pragma unroll
    for (int i = 0; i < C::NA; ++i) {
      const unsigned off = (kc * C::KG + a_grp >= cin_groups) ? OOB : a_off[i];
      ra[i][0] = __builtin_amdgcn_raw_buffer_load_b128(xrs, off, kc * (8 * C::KG * XE), 0);
      if (!XB) ra[i][1] = __builtin_amdgcn_raw_buffer_load_b128(xrs, off + 16, kc * (8 * C::KG * XE), 0);
    }
#pragma unroll
    for (int i = 0; i < C::NBL; ++i) rb[i] = __builtin_amdgcn_raw_buffer_load_b128(wrs, b_off[i], kc * C::KG * a.CoutP * 16, 0);
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < C::NA; ++i)
      As[a_dst[i]] = XB ? __builtin_bit_cast(uint4, ra[i][0]) : pack_bf16x8(__builtin_bit_cast(f32x4, ra[i][0]), __builtin_bit_cast(f32x4, ra[i][1]));
#pragma unroll
    for (int i = 0; i < C::NBL; ++i)
      if (C::B_U4 % C::NT == 0 || tid + i * C::NT < C::B_U4) Bs[tid + i * C::NT] = __builtin_bit_cast(uint4, rb[i]);
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

  // K chunks that hold real channels: the pack is zero padded to CinP = a multiple of 32, a 16-channel-chunk tile (KG = 2) on a layer with
  // Cin <= 16 (conv1_1's 3 -> 8 input channels, the side branches' 16-channel data gradients) used to walk a second chunk of nothing but zeros --
  // half of its staging and MFMA work (round 5; same bits: the dropped products are exact zeros; step level +0.0 .. +0.4 % at batch 12,
  // profiles/r05_ab_small.txt: these launches are bound by their prologue / epilogue latency, not by the matrix pipe)
  const int nchunks = (a.Cin + 8 * C::KG - 1) / (8 * C::KG);
#ifdef OSVOS_CONV_PROF
  unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tq, tp = __builtin_amdgcn_s_memtime();
  const unsigned long long t_begin = tp;
#define PROF_MARK(k) do { tq = __builtin_amdgcn_s_memtime(); pt[k] += tq - tp; tp = tq; } while (0)
#else
#define PROF_MARK(k) do { } while (0)
#endif
  load_chunk(0);
  PROF_MARK(0);
  for (int kc = 0; kc < nchunks; ++kc) {
    __syncthreads();                     // every wave is done with the previous chunk's tiles
    PROF_MARK(1);
#ifdef OSVOS_CONV_PROF
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    PROF_MARK(2);
#endif
    store_chunk();
    PROF_MARK(3);
    __syncthreads();
    PROF_MARK(4);
    if (kc + 1 < nchunks) load_chunk(kc + 1);      // in flight during the MFMAs below
    PROF_MARK(5);
    // 9 x KS (tap, k-step) stages, software pipelined: operands of stage s+1 are requested before the MFMAs of stage s issue;
    // sched_barrier pins that order.  (Measured and removed, see DESIGN.md 3.4: re-using A fragments across tap rows -- 40 % fewer
    // LDS reads -- and issuing the next chunk's loads one per MFMA inside this loop were both no faster.)
    uint4 fa[1 + C::PIPE][C::WM], fb[1 + C::PIPE][C::WN];
    auto ldfrag = [&](int st, int set) {
      const int tap = st / C::KS, ks = st % C::KS;
      const int r = tap / 3, s = tap % 3;
#pragma unroll
      for (int mi = 0; mi < C::WM; ++mi) fa[set][mi] = As[a_idx[mi] + 2 * ks * C::PLANE + r * C::PITCH + s];
#pragma unroll
      for (int ni = 0; ni < C::WN; ++ni) fb[set][ni] = Bs[b_idx + (tap * C::KG + 2 * ks) * C::BN + ni * 32];
    };
    if (C::PIPE) ldfrag(0, 0);
#pragma unroll
    for (int st = 0; st < 9 * C::KS; ++st) {
      const int cur = C::PIPE ? (st & 1) : 0;
      if (C::PIPE) {
        if (st + 1 < 9 * C::KS) ldfrag(st + 1, (st + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
      } else {
        ldfrag(st, 0);     // two waves per SIMD cover each other's LDS latency, registers are the scarce resource
      }
#pragma unroll
      for (int mi = 0; mi < C::WM; ++mi)
#pragma unroll
        for (int ni = 0; ni < C::WN; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb[cur][ni]),
                                                                __builtin_bit_cast(bf16x8_t, fa[cur][mi]), acc[mi][ni], 0, 0, 0);
      if (C::PIPE) __builtin_amdgcn_sched_barrier(0);
    }
    PROF_MARK(6);
  }

  // ---- epilogue.  The weight fragment is the FIRST MFMA operand, so D = [cout rows][pixel columns]: lane (li, lh)
  // holds pixel li of its M block and, per accumulator, couts 8 q + 4 lh + (0..3) in registers 4q..4q+3 -- four
  // consecutive NHWC channels = one 16-byte store.  Stores (and bias / mask loads) are raw buffer accesses whose
  // offset is pushed out of range for pixels outside the image and couts past Cout: no branches, no 64-bit
  // address arithmetic (the former per-element dword epilogue took 40 % of a wave's lifetime at Cin = 256).
  const size_t img_elems = (size_t)a.H * a.W * a.y_cs;
  if ((a.Cout & 3) == 0 && (a.y_cs & 3) == 0) {
    // (a descriptor with num_records = 0 drops every access: absent tensors need no branch)
    void* const anyp = const_cast<uint4*>(a.wpk);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(a.y != nullptr ? (void*)(a.y + n * img_elems) : anyp, 0,
                                                                         a.y != nullptr ? (int)(img_elems * 4) : 0, 0x00020000);
    const int msz = a.mask_bf16 ? 2 : 4;
    const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(
        a.mask != nullptr ? (void*)(reinterpret_cast<char*>(const_cast<void*>(a.mask)) + n * img_elems * msz) : anyp, 0,
        a.mask != nullptr ? (int)(img_elems * msz) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(a.bias != nullptr ? (void*)const_cast<float*>(a.bias) : anyp, 0,
                                                                         a.bias != nullptr ? a.Cout * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(a.ybf != nullptr ? (void*)(a.ybf + n * img_elems) : anyp, 0,
                                                                         a.ybf != nullptr ? (int)(img_elems * 2) : 0, 0x00020000);
    const bool pool_fwd = a.pooled != nullptr;
    const int PHo = (a.H + 1) / 2, PWo = (a.W + 1) / 2;
    const size_t poimg = (size_t)PHo * PWo * a.y_cs;
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(pool_fwd ? (void*)(a.pooled + n * poimg) : anyp, 0, pool_fwd ? (int)(poimg * 2) : 0, 0x00020000);
    const bool pool_code = pool_fwd && a.pool_code != nullptr;
    const __amdgpu_buffer_rsrc_t pcrs = __builtin_amdgcn_make_buffer_rsrc(pool_code ? (void*)(a.pool_code + n * poimg) : anyp, 0, pool_code ? (int)poimg : 0, 0x00020000);
    // one-bit masks (maskbits.h): words per pixel = y_cs / 32
    const int bw = a.y_cs >> 5;
    const size_t img_words = (size_t)a.H * a.W * bw;
    const __amdgpu_buffer_rsrc_t mbrs = __builtin_amdgcn_make_buffer_rsrc(a.mask_bits != nullptr ? (void*)const_cast<unsigned*>(a.mask_bits + n * img_words) : anyp, 0,
                                                                          a.mask_bits != nullptr ? (int)(img_words * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t ybrs = __builtin_amdgcn_make_buffer_rsrc(a.y_bits != nullptr ? (void*)(a.y_bits + n * img_words) : anyp, 0,
                                                                          a.y_bits != nullptr ? (int)(img_words * 4) : 0, 0x00020000);
#pragma unroll
    for (int ni = 0; ni < C::WN; ++ni) {
      const int cb = co0 + (wn * C::WN + ni) * 32 + 4 * lh;
      f32x4 bv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) bv[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, (cb + 8 * q) * 4, 0, 0));
      u32x4 keep[C::WM][2];      // (pool forward only: the packed bf16 results of this wave, zero outside the image)
#pragma unroll
      for (int mi = 0; mi < C::WM; ++mi) {
        const int mb = wm * C::WM + mi;
        const int oy = y0 + (mb / C::TBX) * C::RBH + li / C::RBW;
        const int ox = x0 + (mb % C::TBX) * C::RBW + li % C::RBW;
        const unsigned pix = (oy < a.H && ox < a.W) ? (unsigned)((oy * a.W + ox) * a.y_cs) * 4u : OOB;
        uint2 hb[4];               // bf16 copy: couts 8q + 4lh + (0..3) of this lane, packed
        const unsigned bitoff = (pix != OOB && cb - 4 * lh < a.Cout) ? (unsigned)(oy * a.W + ox) * (unsigned)(bw * 4) + (unsigned)((cb - 4 * lh) >> 5) * 4u : OOB;
        const unsigned mword = mb_load(mbrs, bitoff);      // (an absent tensor's descriptor drops the access)
        unsigned ybits = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int co = cb + 8 * q;
          const unsigned off = co < a.Cout ? pix + (unsigned)co * 4u : OOB;
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = acc[mi][ni][4 * q + e] + bv[q][e];
            if (a.relu) v[e] = v[e] > 0.f ? v[e] : 0.f;
          }
          if (a.mask_bits != nullptr) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = mb_test(mword, q, lh, e) ? v[e] : 0.f;
          } else if (a.mask != nullptr) {
            if (a.mask_bf16) {       // post-ReLU activations stored as bf16: > 0 <=> the 16-bit pattern is a positive integer
              typedef short s16x4 __attribute__((ext_vector_type(4)));
              const s16x4 m = __builtin_bit_cast(s16x4, __builtin_amdgcn_raw_buffer_load_b64(mrs, off >> 1, 0, 0));
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = m[e] > 0 ? v[e] : 0.f;
            } else {
              const f32x4 m = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(mrs, off, 0, 0));
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = m[e] > 0.f ? v[e] : 0.f;
            }
          }
          if (a.y != nullptr) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, off, 0, 0);
          bf16x4_t h;
          h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
          hb[q] = __builtin_bit_cast(uint2, h);
          ybits |= a.ybf != nullptr ? mb_bits_bf16(hb[q], q) : mb_bits_f32(v, q);      // (the stored value decides: bf16 when that is what is kept)
        }
        if (a.y_bits != nullptr) mb_store(ybrs, bitoff, ybits, lh);
        if (a.ybf != nullptr) {
          // lanes l and l+32 hold the two halves of every 8-cout group: v_permlane32_swap trades the odd-q quad of the
          // lower lane for the even-q quad of the upper one, after which each lane owns 8 consecutive couts
          // (16 p + 8 lh .. + 7) = one 16-byte store instead of two 8-byte ones
#pragma unroll
          for (int pq = 0; pq < 2; ++pq) {
            const auto sx = __builtin_amdgcn_permlane32_swap(hb[2 * pq].x, hb[2 * pq + 1].x, false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(hb[2 * pq].y, hb[2 * pq + 1].y, false, false);
            const u32x4 o = {sx[0], sy[0], sx[1], sy[1]};
            const int co = co0 + (wn * C::WN + ni) * 32 + 16 * pq + 8 * lh;
            const unsigned off = co < a.Cout ? (pix >> 1) + (unsigned)co * 2u : OOB;       // (a.Cout % 8 == 0 checked on the host)
            __builtin_amdgcn_raw_buffer_store_b128(o, hrs, off, 0, 0);
            if (pool_fwd) keep[mi][pq] = pix != OOB ? o : u32x4{0, 0, 0, 0};
          }
        }
      }
      if (pool_fwd) {
        // fused forward pool (this launch is the last convolution of a stage): max over the 2 x 2 window of the packed bf16 results.  Post-ReLU
        // values are >= 0, so positions outside the image count as 0 and 16-bit UNSIGNED integer max is the bf16 max.
        // windows: RBW 32 -- M blocks 2j, 2j+1 of this wave are the two rows, lane ^ 1 the neighbouring column;
        //          RBW 16 -- an M block holds both rows (lanes li and li ^ 16), lane ^ 1 the neighbouring column
        typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
        constexpr int NP = (C::RBW == 32) ? C::WM / 2 : C::WM;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
          const int mb = wm * C::WM + (C::RBW == 32 ? 2 * j : j);
          const int oy = y0 + (mb / C::TBX) * C::RBH, ox = x0 + (mb % C::TBX) * C::RBW + li % C::RBW;      // top row of the window pair
          const bool writer = (li & 1) == 0 && ((C::RBW == 32) || li < 16) && oy < a.H && ox < a.W;
          const unsigned ppix = writer ? (unsigned)(((oy >> 1) * PWo + (ox >> 1)) * a.y_cs) * 2u : OOB;
#pragma unroll
          for (int pq = 0; pq < 2; ++pq) {
            // this lane's column: r0 = its upper row, r1 = its lower row (RBW 16: for lanes li >= 16 the other way round -- they never write)
            u32x4 r0, r1;
            if constexpr ((C::RBW == 32)) {
              r0 = keep[2 * j][pq];
              r1 = keep[2 * j + 1][pq];
            } else {
              r0 = keep[j][pq];
              r1 = u32x4{(unsigned)__shfl_xor((int)r0[0], 16, 64), (unsigned)__shfl_xor((int)r0[1], 16, 64), (unsigned)__shfl_xor((int)r0[2], 16, 64),
                         (unsigned)__shfl_xor((int)r0[3], 16, 64)};
            }
            u16x8 m = __builtin_elementwise_max(__builtin_bit_cast(u16x8, r0), __builtin_bit_cast(u16x8, r1));
            const int co = co0 + (wn * C::WN + ni) * 32 + 16 * pq + 8 * lh;
            if (!pool_code) {
              const u32x4 t = __builtin_bit_cast(u32x4, m);
              const u32x4 u = {(unsigned)__shfl_xor((int)t[0], 1, 64), (unsigned)__shfl_xor((int)t[1], 1, 64), (unsigned)__shfl_xor((int)t[2], 1, 64),
                               (unsigned)__shfl_xor((int)t[3], 1, 64)};
              m = __builtin_elementwise_max(m, __builtin_bit_cast(u16x8, u));
            } else {
              // the neighbouring column's two rows arrive separately: the writer needs all four window values for the code byte
              const u32x4 n0 = {(unsigned)__shfl_xor((int)r0[0], 1, 64), (unsigned)__shfl_xor((int)r0[1], 1, 64), (unsigned)__shfl_xor((int)r0[2], 1, 64),
                                (unsigned)__shfl_xor((int)r0[3], 1, 64)};
              const u32x4 n1 = {(unsigned)__shfl_xor((int)r1[0], 1, 64), (unsigned)__shfl_xor((int)r1[1], 1, 64), (unsigned)__shfl_xor((int)r1[2], 1, 64),
                                (unsigned)__shfl_xor((int)r1[3], 1, 64)};
              m = __builtin_elementwise_max(m, __builtin_elementwise_max(__builtin_bit_cast(u16x8, n0), __builtin_bit_cast(u16x8, n1)));
              unsigned cw[2] = {0u, 0u};
#pragma unroll
              for (int e = 0; e < 8; ++e) {      // scan order (0,0) (0,1) (1,0) (1,1), strict >: the first maximum (pool.hip); post-ReLU bf16 compare as u16
                const int sh = 16 * (e & 1);
                const unsigned av = (r0[e >> 1] >> sh) & 0xffffu, bv = (n0[e >> 1] >> sh) & 0xffffu;
                const unsigned cv = (r1[e >> 1] >> sh) & 0xffffu, dv = (n1[e >> 1] >> sh) & 0xffffu;
                unsigned bi = 0u, best = av;
                if (bv > best) { best = bv; bi = 1u; }
                if (cv > best) { best = cv; bi = 2u; }
                if (dv > best) { bi = 3u; }
                const unsigned byte = bi | (av ? 4u : 0u) | (bv ? 8u : 0u) | (cv ? 16u : 0u) | (dv ? 32u : 0u);
                cw[e >> 2] |= byte << (8 * (e & 3));
              }
              typedef unsigned u32x2s __attribute__((ext_vector_type(2)));
              __builtin_amdgcn_raw_buffer_store_b64(u32x2s{cw[0], cw[1]}, pcrs, (co < a.Cout && ppix != OOB) ? (ppix >> 1) + (unsigned)co : OOB, 0, 0);
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, m), prs, (co < a.Cout && ppix != OOB) ? ppix + (unsigned)co * 2u : OOB, 0, 0);
          }
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
            if (a.mask != nullptr) v = reinterpret_cast<const float*>(a.mask)[o] > 0.f ? v : 0.f;
            a.y[o] = v;
          }
        }
      }
  }
#ifdef OSVOS_CONV_PROF
  PROF_MARK(7);
  if (a.prof != nullptr && lane == 0) {
    unsigned long long* q = a.prof + ((size_t)blockIdx.x * (C::NT / 64) + wave) * 10;
    for (int k = 0; k < 8; ++k) q[k] = pt[k];
    q[8] = t_begin;
    q[9] = tp;
  }
#endif
}

template <class C, int XB = 0>
int launch_cfg(const ConvArgsB& a0, hipStream_t stream) {
  static bool attr_set_dev[OSVOS_MAX_DEVICES] = {};      // hipFuncSetAttribute is per device
  bool& attr_set = attr_set_dev[osvos_current_device()];
  if (!attr_set) {
    OSVOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_bf16_kernel<C, XB>),
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
  hipLaunchKernelGGL((conv3x3_bf16_kernel<C, XB>), dim3((unsigned)blocks), dim3(C::NT), C::LDS_BYTES, stream, a);
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
//                  RBW TBX TBY NB WGM WGN PIPE OCC KG
using B8 = CfgB<32, 1, 8, 4, 2, 2, 0, 2, 2>;   // 256 px x 128 co, 4x2 accumulators, 16-channel chunks -> two workgroups per CU (the workhorse)
using B9 = CfgB<32, 1, 8, 2, 2, 2, 1, 3, 2>;   // 256 px x  64 co, 16-channel chunks, three workgroups per CU (conv1_x)
using B10 = CfgB<16, 1, 8, 4, 2, 2, 0, 2, 2>;  // B8 as a 16 x 16 pixel tile: 107- and 54-pixel wide maps lose 12 % to padding instead of 28 %
using B11 = CfgB<16, 1, 8, 2, 2, 2, 1, 1, 4>;  // B1 as a 16 x 16 pixel tile
// (B6 on 16-channel chunks with three workgroups per CU, tried for the K = 64 input gradient: 279 against 276 us at batch 12 -- not kept)
// (tile ids in the round-1 profile files predate a clean-up: 20 -> 8, 23 -> 9, 28 -> 10, 29 -> 11; the 8-wave, row-re-use and
//  interleaved-load variants 8-19 / 21-22 / 24-27 of those files were measured, lost, and are gone)
constexpr int kNumTilesB = 12;
template <class C>
constexpr TileInfoB infoB() { return TileInfoB{C::TW, C::TH, C::BN, C::WM, C::WN, C::LDS_BYTES}; }
const TileInfoB kTilesB[kNumTilesB] = {infoB<B0>(), infoB<B1>(), infoB<B2>(), infoB<B3>(), infoB<B4>(), infoB<B5>(), infoB<B6>(), infoB<B7>(),
                                       infoB<B8>(), infoB<B9>(), infoB<B10>(), infoB<B11>()};

// Measured (profiles/r01_tune_bf16_*.txt): with >= 128 couts the 256 px x 128 co tile on 16-channel chunks (B8: 4x2
// accumulators per wave, two workgroups per CU) moves the fewest bytes per MFMA and wins whenever it yields enough
// workgroups (up to 990 TFLOP/s on conv3_x/conv4_x at batch 12); B1 (256 px x 64 co, 4 accumulators) covers the
// 64-cout layers and the frames that are too small for B8; tiny frames fall back to 128- and 64-pixel tiles.
constexpr int kDmaMinCin = 512, kDmaTile = 32;      // auto rule for the LDS-DMA kernel (OSVOS_DMA_MIN_CIN / OSVOS_DMA_TILE override)

int pick_tile_b(int N, int H, int W, int CoutP, int Cin) {
  if (CoutP <= 32) return 6;
  // one 16-channel K chunk (the side branches' data gradients, 16 -> C): all prologue and epilogue -- three small workgroups per CU cover each
  // other's phases (B9: 166 / 80 / 52 us against 215 / 90 / 57 for the automatic choice at batch 12, tools/tune_skinny_bf16.py)
  if (Cin <= 16) return 9;
  // ... and the Cin = 64 layers (conv1_2, conv2_1; their data gradients): K = 576 is four such chunks, the launch is as much prologue / epilogue and
  // HBM stream as matrix work -- the same small tile, three workgroups per CU, reads 2-5 % faster than the 32-channel-chunk tile at batch 12
  // (tools/tune_conv.py) and the step +0.7-1.2 % (1136.3 / 1134.2 -> 1143.8 / 1147.5 frames/s, profiles/r05_ab_small.txt; Cin <= 128: +0.1-0.8 %)
  // (guarded in round 6, ADVICE r05: the rule was measured at 854x480 batch 12 only -- tiny frames and crops keep the size-aware choice below, where
  //  a 107- or 54-pixel wide map loses up to 28 % of a 32 x 8 tile to padding)
  if (Cin <= 64 && CoutP <= 128 && (long)N * ceil_div(H, kTilesB[9].th) * ceil_div(W, kTilesB[9].tw) * ceil_div(CoutP, kTilesB[9].bn) >= 400 &&
      (long)ceil_div(H, 8) * 8 * ceil_div(W, 32) * 32 * 100 <= (long)H * W * 115)
    return 9;
  const int order[] = {8, 1, 5, 7};
  for (int k = 0; k < 4; ++k) {
    const TileInfoB& t = kTilesB[order[k]];
    if (t.bn > CoutP) continue;
    const long tiles = (long)N * ceil_div(H, t.th) * ceil_div(W, t.tw) * ceil_div(CoutP, t.bn);
    if (tiles >= 400 || k == 3) {
      if (order[k] == 8) {      // same tile as 16 x 16 pixels when that pads the frame less (107-pixel wide conv4_x: 12 % vs 28 %)
        const long p20 = (long)ceil_div(H, 8) * 8 * ceil_div(W, 32) * 32, p28 = (long)ceil_div(H, 16) * 16 * ceil_div(W, 16) * 16;
        if (p28 * 100 < p20 * 92) return 10;
      }
      return order[k];
    }
  }
  return 7;
}

// wpk[((tap*CG + cg)*CoutP + co)*8 + e] = bf16(W[co][8cg+e][tap])   (zero padded)
// every layer's bf16 packs, forward and data-gradient form, in ONE launch (round 5 prep; the f32x3 twin is pack_x3_multi_kernel): osvos_net_pack
// re-packs 17 filters x 2 forms after every optimizer step -- 34 launches of ~8 us each, one float per thread at a 36-byte stride.  A unit =
// one (8-channel group cg of the reduction dimension, 32 output channels) block: its source values are whole contiguous runs of the OIHW
// filter (forward: 72 floats per output channel; data gradient: 288 floats per reduction channel), turned through LDS, written as nine
// 512-byte runs.  Reduction channels are padded to a multiple of 32 and output channels to a multiple of 32 with zeros, like the single packs.
struct PackB16Table {
  const float* w[OSVOS_PACK_MAX];
  bf16_t* dst[OSVOS_PACK_MAX];
  int Cout[OSVOS_PACK_MAX], Cin[OSVOS_PACK_MAX], dgrad[OSVOS_PACK_MAX];
  long start[OSVOS_PACK_MAX + 1];      // in (cg, 32-channel) units
  int n;
};

__global__ __launch_bounds__(256) void pack_bf16_multi_kernel(PackB16Table t) {
  constexpr int ROW = 8 * 9 + 1;
  __shared__ float tile[32 * ROW];
  for (long blk = blockIdx.x; blk < t.start[t.n]; blk += gridDim.x) {
    int k = 0;
    while (blk >= t.start[k + 1]) ++k;
    const int dgrad = t.dgrad[k], Cout = t.Cout[k], Cin = t.Cin[k];
    const int K = dgrad ? Cout : Cin, M = dgrad ? Cin : Cout;       // reduction / output channels of the convolution this pack feeds
    const int KP = (K + 31) / 32 * 32, MP = (M + 31) / 32 * 32, CG = KP / 8;
    const int u = (int)(blk - t.start[k]);
    const int cg = u % CG, m0 = (u / CG) * 32;
    const float* __restrict__ w = t.w[k];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 9; ++it) {
      const int L = it * 256 + (int)threadIdx.x;
      int ml, e, tap;
      float v = 0.f;
      if (dgrad) {                                           // row = reduction channel kk = 8 cg + e (a Cout index): [m0 .. m0 + 31][9] contiguous
        e = L / 288;
        const int j = L % 288;
        ml = j / 9;
        tap = 8 - j % 9;
        if (m0 + ml < M && cg * 8 + e < K) v = w[((long)(cg * 8 + e) * Cin + m0) * 9 + j];
      } else {                                               // row = output channel m0 + ml: [8 cg .. 8 cg + 7][9] contiguous (ragged at K = 3)
        ml = L / 72;
        const int j = L % 72;
        e = j / 9;
        tap = j % 9;
        if (m0 + ml < M && cg * 8 + e < K) v = w[((long)(m0 + ml) * Cin + cg * 8) * 9 + j];
      }
      tile[ml * ROW + e * 9 + tap] = v;
    }
    __syncthreads();
    const int ml = (int)threadIdx.x >> 3, e = (int)threadIdx.x & 7;
    bf16_t* d = t.dst[k];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) d[(((long)tap * CG + cg) * MP + m0 + ml) * 8 + e] = f32_to_bf16(tile[ml * ROW + e * 9 + tap]);
  }
}

}  // namespace

int osvos_conv3x3_bf16mfma_num_tiles(void) { return kNumTilesB; }

int osvos_pack_fwd_bf16(const float* w, void* wpk, int Cout, int Cin, hipStream_t stream) {
  OSVOS_ARG_CHECK(w && wpk && Cout > 0 && Cin > 0, "pack_fwd bf16: bad arguments");
  const float* ws[1] = {w};
  void* dsts[1] = {wpk};
  const int co[1] = {Cout}, ci[1] = {Cin}, dg[1] = {0};
  return osvos_pack_bf16_multi(ws, dsts, co, ci, dg, 1, stream);
}

int osvos_pack_dgrad_bf16(const float* w, void* wpk, int Cout, int Cin, hipStream_t stream) {
  OSVOS_ARG_CHECK(w && wpk && Cout > 0 && Cin > 0, "pack_dgrad bf16: bad arguments");
  const float* ws[1] = {w};
  void* dsts[1] = {wpk};
  const int co[1] = {Cout}, ci[1] = {Cin}, dg[1] = {1};
  return osvos_pack_bf16_multi(ws, dsts, co, ci, dg, 1, stream);
}

// n bf16 packs (n <= OSVOS_PACK_MAX) in one launch: ws[k] OIHW fp32 [Couts[k]][Cins[k]][3][3] -> dsts[k] (osvos_pack_fwd_bf16 layout; dgrads[k] != 0:
// osvos_pack_dgrad_bf16 layout)
int osvos_pack_bf16_multi(const float* const* ws, void* const* dsts, const int* Couts, const int* Cins, const int* dgrads, int n, hipStream_t stream) {
  OSVOS_ARG_CHECK(ws && dsts && Couts && Cins && dgrads && n >= 0 && n <= OSVOS_PACK_MAX, "pack_bf16_multi: bad table (n = %d)", n);
  if (n == 0) return 0;
  PackB16Table t;
  t.n = n;
  t.start[0] = 0;
  for (int k = 0; k < n; ++k) {
    const int K = dgrads[k] ? Couts[k] : Cins[k], M = dgrads[k] ? Cins[k] : Couts[k];
    OSVOS_ARG_CHECK(ws[k] && dsts[k] && K > 0 && M > 0, "pack_bf16_multi: entry %d (K = %d, M = %d)", k, K, M);
    t.w[k] = ws[k]; t.dst[k] = reinterpret_cast<bf16_t*>(dsts[k]); t.Cout[k] = Couts[k]; t.Cin[k] = Cins[k]; t.dgrad[k] = dgrads[k] ? 1 : 0;
    t.start[k + 1] = t.start[k] + (long)(((K + 31) / 32 * 32) / 8) * (osvos_cout_pad(M) / 32);
  }
  const long blocks = t.start[n] < 8192 ? t.start[n] : 8192;
  hipLaunchKernelGGL(pack_bf16_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, t);
  OSVOS_LAUNCH_CHECK();
  return 0;
}

// phase-counter hook of tools/conv_phase_probe.py: exists only in probe builds (make EXTRA=-DOSVOS_CONV_PROF); the shipped library has no
// process-global device pointer behind its re-entrant ABI
#ifdef OSVOS_CONV_PROF
static unsigned long long* g_conv_prof = nullptr;
extern "C" void osvos_debug_set_conv_prof(void* p) { g_conv_prof = (unsigned long long*)p; }
#define OSVOS_CONV_PROF_PTR g_conv_prof
#else
#define OSVOS_CONV_PROF_PTR nullptr
#endif

// x fp32 NHWC (stride Cin, multiple of 8), wpk from osvos_pack_{fwd,dgrad}_bf16 with the same Cin/Cout roles
// xb = 0: x fp32, xb = 1: x bf16; ybf (optional) receives a bf16 copy of y
int osvos_conv3x3_bf16mfma_io(const void* x, int xb, const void* wpk, const float* bias, const void* mask, int mask_bf16, float* y, void* ybf,
                              int N, int H, int W, int Cin, int Cout, int y_cs, int relu, int tile, hipStream_t stream) {
  return osvos_conv3x3_bf16mfma_bits(x, xb, wpk, bias, mask, mask_bf16, nullptr, y, ybf, nullptr, nullptr, N, H, W, Cin, Cout, y_cs, relu, tile, stream, nullptr);
}

// mask_bits: the ReLU mask as one bit per element (maskbits.h; takes precedence over `mask`); y_bits: sign bits of the result, written beside it
// pooled_bf16 (optional; needs ybf, ReLU, a dense result with Cout % 8 == 0): maxpool2x2 (ceil mode) of the bf16 result, written by the same launch
// ([N][ceil(H/2)][ceil(W/2)][Cout]; the 8 x 8-pixel tile cannot hold whole windows per wave: there the pooling kernel is launched behind the convolution)
// pool_code (optional, with pooled_bf16): [N][ceil(H/2)][ceil(W/2)][Cout] bytes for osvos_maxpool2x2_bwd_bf16_code; written whichever kernel runs
int osvos_conv3x3_bf16mfma_bits(const void* x, int xb, const void* wpk, const float* bias, const void* mask, int mask_bf16, const unsigned* mask_bits,
                                float* y, void* ybf, unsigned* y_bits, void* pooled_bf16, int N, int H, int W, int Cin, int Cout, int y_cs, int relu, int tile,
                                hipStream_t stream, void* pool_code) {
  OSVOS_ARG_CHECK(pool_code == nullptr || pooled_bf16 != nullptr, "conv3x3 bf16: pool code bytes without a pooled result");
  OSVOS_ARG_CHECK(pooled_bf16 == nullptr || (ybf != nullptr && relu && mask == nullptr && mask_bits == nullptr && Cout % 8 == 0 && y_cs == Cout),
                  "conv3x3 bf16: the fused forward pool needs a bf16 result, ReLU, no mask and a dense Cout %% 8 == 0 (Cout %d, stride %d)", Cout, y_cs);
  OSVOS_ARG_CHECK(x && wpk && (y || ybf), "conv3x3 bf16: null pointer");
  OSVOS_ARG_CHECK((mask_bits == nullptr && y_bits == nullptr) || (Cout % 32 == 0 && y_cs == Cout), "conv3x3 bf16: one-bit masks need Cout %% 32 == 0 and a dense result (Cout %d, stride %d)", Cout, y_cs);
  OSVOS_ARG_CHECK((Cout % 4 == 0 && y_cs % 4 == 0) || (y != nullptr && !(mask && mask_bf16)),
                  "conv3x3 bf16: ragged channel counts (Cout %d, stride %d) support fp32 outputs and masks only", Cout, y_cs);
  OSVOS_ARG_CHECK(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv3x3 bf16: bad shape");
  OSVOS_ARG_CHECK(Cin % 8 == 0, "conv3x3 bf16: Cin (%d) must be a multiple of 8", Cin);
  OSVOS_ARG_CHECK(y_cs >= Cout, "conv3x3 bf16: y channel stride %d < Cout %d", y_cs, Cout);
  OSVOS_ARG_CHECK((long)H * W * Cin < (1L << 29), "conv3x3 bf16: image too large for 31-bit byte offsets");
  OSVOS_ARG_CHECK((long)H * W * y_cs < (1L << 29), "conv3x3 bf16: output image too large for 31-bit byte offsets");
  OSVOS_ARG_CHECK(ybf == nullptr || (Cout % 8 == 0 && y_cs % 8 == 0), "conv3x3 bf16: the bf16 output copy needs Cout and y_cs multiples of 8");
  ConvArgsB a;
  a.x = x; a.wpk = reinterpret_cast<const uint4*>(wpk); a.bias = bias; a.mask = mask; a.mask_bf16 = mask_bf16 ? 1 : 0; a.y = y;
  a.ybf = reinterpret_cast<bf16_t*>(ybf);
  a.mask_bits = mask_bits; a.y_bits = y_bits; a.pooled = reinterpret_cast<bf16_t*>(pooled_bf16);
  a.pool_code = reinterpret_cast<unsigned char*>(pool_code);
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.CinP = (Cin + 31) / 32 * 32; a.Cout = Cout; a.CoutP = osvos_cout_pad(Cout); a.y_cs = y_cs;
  a.relu = relu;
  a.prof = OSVOS_CONV_PROF_PTR;
  if (tile < 0) {
    OSVOS_ENV_INT(env_tile, "OSVOS_CONV_TILE_BF16", -1);
    const bool env = env_tile >= 0;
    tile = env ? env_tile : pick_tile_b(N, H, W, a.CoutP, Cin);
    if (!env && !xb && tile == 10) tile = 8;     // (the 16 x 16 form spills with fp32 staging registers)
    // bf16 activations, deep layers (K = 9 x 512): the LDS-DMA staged 512 px x 128 co kernel wins when it still fills the chip
    // (conv4_x 0.355 -> 0.331 ms, conv5_x 0.117 -> 0.098 ms at batch 12)
    static const int dma_min_cin = getenv("OSVOS_DMA_MIN_CIN") ? atoi(getenv("OSVOS_DMA_MIN_CIN")) : kDmaMinCin;
    static const int dma_tile = getenv("OSVOS_DMA_TILE") ? atoi(getenv("OSVOS_DMA_TILE")) : kDmaTile;
    // (grid threshold 256 -> 128 in round 5: at batch 12 conv5_x's 192 DMA workgroups still beat the register-staged tile, 0.100-0.104 vs 0.108 ms per
    //  launch and +0.4-0.8 % on configs[2], profiles/r05_ab_small.txt; the persistent DMA tile 35 and "DMA only for Cout >= 512" measured -0.6 % / +0.2 %)
    if (!env && xb && Cin >= dma_min_cin && a.CoutP >= 128 && osvos_conv3x3_bf16_dma_applicable(Cin, Cout, y_cs) &&
        (long)N * ceil_div(H, 16) * ceil_div(W, 32) * ceil_div(a.CoutP, 128) >= 128)
      tile = dma_tile;
    // Cin = 64, bf16 in / out, FORWARD launches with enough 512-pixel tiles to give every CU several: the persistent resident-filter kernel with the
    // deferred + skewed packed epilogue (conv3x3_bf16_p64.hip; round 6): 6-10 % faster at op level at batch 12 (conv1_2 + pool 0.524 vs 0.582 ms,
    // conv2_1 0.207 vs 0.226), step level within noise (profiles/r06_p64_diagnosis.txt).  NOT the data gradient (17 % faster alone, but one 160 KB / 256-VGPR
    // workgroup per CU cannot share a CU with the weight-gradient stream the way tile 9's three small workgroups do).  OSVOS_P64_MIN_TILES: tiles per
    // launch from which it is taken (0 = never); OSVOS_P64_DGRAD=1 takes it for masked launches too.
    OSVOS_ENV_INT(p64_min, "OSVOS_P64_MIN_TILES", 1024);
    OSVOS_ENV_INT(p64_dgrad, "OSVOS_P64_DGRAD", 0);
    if (!env && xb && p64_min > 0 && (p64_dgrad || mask_bits == nullptr) && (long)N * ceil_div(H, 16) * ceil_div(W, 32) * ceil_div(Cout, 64) >= p64_min &&
        osvos_conv3x3_bf16_p64_applicable(Cin, Cout, y_cs, y != nullptr, mask != nullptr, mask_bits != nullptr, y_bits != nullptr, pooled_bf16 != nullptr, relu))
      tile = 38;
    // (round 5 re-check at step level, configs[2]: always / never / 3x / 10x this threshold all within +-0.3 % -- unlike the f32x3 rule)
    if (!env && (double)H * W * Cin * 4 > 9.0 * Cin * a.CoutP * 2) tile += 100;
  }
  a.map = tile >= 100 ? 1 : 0;
  tile %= 100;
  if (xb && tile == 38) {      // Cin = 64: persistent, resident filter, deferred + skewed packed epilogue (conv3x3_bf16_p64.hip)
    OSVOS_ARG_CHECK(osvos_conv3x3_bf16_p64_applicable(Cin, Cout, y_cs, y != nullptr, mask != nullptr, mask_bits != nullptr, y_bits != nullptr, pooled_bf16 != nullptr, relu),
                    "conv3x3 bf16: tile 38 needs Cin = 64, bf16 in / out only, no full-tensor mask, sign bits only with ReLU (Cin %d, Cout %d)", Cin, Cout);
    return osvos_conv3x3_bf16_p64(x, wpk, bias, mask_bits, ybf, y_bits, pooled_bf16, pool_code, N, H, W, Cout, y_cs, relu, a.map, stream);
  }
  if (xb && tile >= 30 && tile <= 37) {      // LDS-DMA staged kernel (its fused pool writes the code bytes too: round 6)
    OSVOS_ARG_CHECK(osvos_conv3x3_bf16_dma_applicable(Cin, Cout, y_cs), "conv3x3 bf16: tile %d (DMA staging) needs Cin %% 16 == 0, Cout, y_cs %% 8 == 0", tile);
    OSVOS_ARG_CHECK(tile < 36 || Cin == 64, "conv3x3 bf16: tile %d (resident filter) is built for Cin = 64 (got %d)", tile, Cin);
    return osvos_conv3x3_bf16_dma(x, wpk, bias, mask, mask_bf16, mask_bits, y, ybf, y_bits, pooled_bf16, N, H, W, Cin, Cout, y_cs, relu, tile - 30, a.map, stream, pool_code);
  }
  if (a.pooled != nullptr && !(xb && tile != 7 && tile >= 0 && tile < kNumTilesB)) {      // a tile whose waves do not hold whole windows: separate pooling launch
    a.pooled = nullptr;
    const int rc = osvos_conv3x3_bf16mfma_bits(x, xb, wpk, bias, mask, mask_bf16, mask_bits, y, ybf, y_bits, nullptr, N, H, W, Cin, Cout, y_cs, relu, tile + 100 * a.map, stream,
                                               nullptr);
    return rc ? rc : osvos_maxpool2x2_bf16_code(ybf, pooled_bf16, pool_code, N, H, W, Cout, stream);
  }
  if (xb) {
    switch (tile) {
      case 0: return launch_cfg<B0, 1>(a, stream);
      case 1: return launch_cfg<B1, 1>(a, stream);
      case 2: return launch_cfg<B2, 1>(a, stream);
      case 3: return launch_cfg<B3, 1>(a, stream);
      case 4: return launch_cfg<B4, 1>(a, stream);
      case 5: return launch_cfg<B5, 1>(a, stream);
      case 6: return launch_cfg<B6, 1>(a, stream);
      case 7: return launch_cfg<B7, 1>(a, stream);
      case 8: return launch_cfg<B8, 1>(a, stream);
      case 9: return launch_cfg<B9, 1>(a, stream);
      case 10: return launch_cfg<B10, 1>(a, stream);
      case 11: return launch_cfg<B11, 1>(a, stream);
      default: osvos_set_error("conv3x3 bf16: unknown tile config %d", tile); return -1;
    }
  }
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
    case 10: return launch_cfg<B10>(a, stream);
    case 11: return launch_cfg<B11>(a, stream);
    default: osvos_set_error("conv3x3 bf16: unknown tile config %d", tile); return -1;
  }
}

int osvos_conv3x3_bf16mfma_xb_tiles(int* tiles, int max) {      // tile ids built for bf16 activations
  static const int t[] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 30, 31, 32, 33, 34, 35, 36, 37};      // (36, 37: Cin = 64 only)
  int n = 0;
  for (; n < (int)(sizeof(t) / sizeof(t[0])) && n < max; ++n) tiles[n] = t[n];
  return n;
}

// x fp32 NHWC (stride Cin, multiple of 8), wpk from osvos_pack_{fwd,dgrad}_bf16 with the same Cin/Cout roles
int osvos_conv3x3_bf16mfma(const float* x, const void* wpk, const float* bias, const float* mask, float* y,
                           int N, int H, int W, int Cin, int Cout, int y_cs, int relu, int tile, hipStream_t stream) {
  return osvos_conv3x3_bf16mfma_io(x, 0, wpk, bias, mask, 0, y, nullptr, N, H, W, Cin, Cout, y_cs, relu, tile, stream);
}

#ifdef OSVOS_CONV_PROF   // C entry points of the scratch library tools/conv_phase_probe.py builds from this file alone
extern "C" int osvos_prof_pack_fwd_bf16(const float* w, void* wpk, int Cout, int Cin) { return osvos_pack_fwd_bf16(w, wpk, Cout, Cin, nullptr); }
extern "C" int osvos_prof_conv3x3_bf16mfma(const float* x, const void* wpk, float* y, int N, int H, int W, int Cin, int Cout, int tile) {
  return osvos_conv3x3_bf16mfma(x, wpk, nullptr, nullptr, y, N, H, W, Cin, Cout, Cout, 1, tile, nullptr);
}
#endif
