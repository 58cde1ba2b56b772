// Internal helpers shared by the HIP translation units of libosvos_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/osvos_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short bf16_t;   // raw bfloat16 bits
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

void osvos_set_error(const char* fmt, ...);
int osvos_wgrad_phase();            // 0 both, 1 partial slabs only, 2 slab reduce only (errors.cpp)
void osvos_wgrad_set_phase(int p);
int osvos_x3_pieces();             // bf16 pieces per operand of the f32x3 kernels on this thread: 3 (default) or 2 (errors.cpp)

#define OSVOS_ARG_CHECK(cond, ...)                   \
  do {                                               \
    if (!(cond)) {                                   \
      osvos_set_error(__VA_ARGS__);                  \
      return -1;                                     \
    }                                                \
  } while (0)

#define OSVOS_HIP_CHECK(expr)                                                        \
  do {                                                                               \
    hipError_t e_ = (expr);                                                          \
    if (e_ != hipSuccess) {                                                          \
      osvos_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
      return (int)e_;                                                                \
    }                                                                                \
  } while (0)

#define OSVOS_LAUNCH_CHECK() OSVOS_HIP_CHECK(hipGetLastError())

// per-device one-time state (kernel attributes): a process may drive more than one GPU
#define OSVOS_MAX_DEVICES 64
static inline int osvos_current_device() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= OSVOS_MAX_DEVICES) d = 0;
  return d;
}
// compute units of the current device (cached per device)
static inline int osvos_cu_count() {
  static int cus[OSVOS_MAX_DEVICES] = {};
  int& c = cus[osvos_current_device()];
  if (c == 0) {
    int v = 0;
    c = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, osvos_current_device()) == hipSuccess && v > 0) ? v : 256;
  }
  return c;
}
// integer environment knob, read once per process (tuning / test switches must not cost a getenv per launch)
#define OSVOS_ENV_INT(var, name, dflt) static const int var = [] { const char* e_ = getenv(name); return e_ ? atoi(e_) : (dflt); }()

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

// bf16 <-> f32 (round to nearest even), device + host
__host__ __device__ inline float bf16_to_f32(bf16_t v) {
  union { uint32_t u; float f; } c;
  c.u = (uint32_t)v << 16;
  return c.f;
}
__host__ __device__ inline bf16_t f32_to_bf16(float f) {
  union { uint32_t u; float f; } c;
  c.f = f;
  if ((c.u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((c.u >> 16) | 0x40);  // NaN
  uint32_t r = c.u + 0x7fffu + ((c.u >> 16) & 1u);
  return (bf16_t)(r >> 16);
}

// wave (64 lanes) sum reductions
__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// channels per 16-byte group / padded channel counts for a dtype
static inline int osvos_group(int dtype) { return dtype == OSVOS_BF16 ? 8 : 4; }
static inline int osvos_cin_pad(int cin, int dtype) { int g2 = 2 * osvos_group(dtype); return (cin + g2 - 1) / g2 * g2; }
static inline int osvos_cout_pad(int cout) { return (cout + 31) / 32 * 32; }
static inline size_t osvos_elem(int dtype) { return dtype == OSVOS_BF16 ? 2 : 4; }   // activation element size
static inline bool osvos_dtype_built(int dtype) { return dtype == OSVOS_F32 || dtype == OSVOS_F32_BF16MFMA || dtype == OSVOS_F32_X3; }
