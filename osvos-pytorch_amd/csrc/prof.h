#pragma once
#include "common.h"

#define OSVOS_PROF_NCAT 4
enum { OSVOS_PROF_CONV_FWD = 0, OSVOS_PROF_CONV_BWD = 1, OSVOS_PROF_SPARE2 = 2, OSVOS_PROF_OTHER = 3 };

bool osvos_prof_on();
void osvos_prof_begin(int cat, double flops, hipStream_t stream);
void osvos_prof_end(hipStream_t stream);

struct ProfScope {
  hipStream_t s;
  bool on;
  ProfScope(int cat, double flops, hipStream_t stream) : s(stream), on(osvos_prof_on()) {
    if (on) osvos_prof_begin(cat, flops, s);
  }
  ~ProfScope() {
    if (on) osvos_prof_end(s);
  }
};
