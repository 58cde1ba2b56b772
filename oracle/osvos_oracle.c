/* ORACLE (test infrastructure only): plain-C restatement of the OSVOS hot path (see
 * osvos_oracle_impl.h for the reference citations).  Built by oracle/Makefile into
 * oracle/libosvos_oracle.so; loaded with ctypes by oracle/c_oracle.py.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define REAL float
#define SUF(x) x##_f32
#include "osvos_oracle_impl.h"
#undef REAL
#undef SUF

#define REAL double
#define SUF(x) x##_f64
#include "osvos_oracle_impl.h"
#undef REAL
#undef SUF

/* bilinear deconv filter, osvos_layers.py:59-67 */
void osvos_oracle_upsample_filt(int size, double* out) {
  int factor = (size + 1) / 2;
  double center = (size % 2 == 1) ? factor - 1 : factor - 0.5;
  for (int i = 0; i < size; ++i)
    for (int j = 0; j < size; ++j)
      out[i * size + j] = (1 - fabs(i - center) / factor) * (1 - fabs(j - center) / factor);
}

/* crop offsets (top, left) the reference's negative pad removes, osvos_layers.py:52-56 */
void osvos_oracle_crop_offsets(int hin, int win, int h, int w, int* top, int* left) {
  *top = (hin - h) / 2;
  *left = (win - w) / 2;
}
