// Per-thread error string + version (libosvos_hip.so).
#include "common.h"
#include <stdlib.h>

static thread_local char g_err[512] = "";

void osvos_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* osvos_last_error(void) { return g_err; }
extern "C" int osvos_version(void) { return OSVOS_ABI_VERSION; }

// Which half of a weight-gradient call to enqueue (per host thread): 0 = partial slabs + reduce (default),
// 1 = partial slabs only, 2 = reduce only.  osvos_net_backward uses 1 / 2 to put the bandwidth-bound slab
// reduces on their own stream, off the critical path of the MFMA weight-gradient chain.
static thread_local int g_wgrad_phase = 0;
int osvos_wgrad_phase() { return g_wgrad_phase; }
void osvos_wgrad_set_phase(int p) { g_wgrad_phase = p; }

// pieces per operand of the f32x3 kernels on this host thread: 3 (default: three bf16 pieces, six products, fp32-grade), 2 (two bf16 pieces, three
// products: precision 'fp32x2') or 22 (two FP16 pieces with block exponents, three products: precision 'fp32h2', h2split.h).  Set by osvos_net_forward / osvos_net_backward from OSVOS_FLAG_X3_TWO_PIECES for the duration of the call, or by osvos_set_x3_pieces.
static thread_local int g_x3_pieces = 3;
int osvos_x3_pieces() { return g_x3_pieces; }
extern "C" int osvos_set_x3_pieces(int pieces) {
  OSVOS_ARG_CHECK(pieces == 2 || pieces == 3 || pieces == 22, "set_x3_pieces: %d (2, 3 or 22)", pieces);
  g_x3_pieces = pieces;
  return 0;
}
