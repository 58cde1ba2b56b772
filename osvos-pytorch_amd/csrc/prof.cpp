// Opt-in launch profiler used by bench.py: hipEvent pairs around the kernel launches of
// osvos_net_forward/backward, recorded on the SAME stream the kernels run on, aggregated per
// kernel family.  Off by default (one predictable branch per launch site); the only mutable
// global state in the library and never touched by the hot path unless bench.py turns it on.
#include <vector>

#include "prof.h"

namespace {
struct Rec { hipEvent_t a, b; int cat; double flops; };
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
size_t g_pool_next = 0;
bool g_on = false;
}  // namespace

bool osvos_prof_on() { return g_on; }

void osvos_prof_begin(int cat, double flops, hipStream_t stream) {
  if (!g_on || g_pool_next + 2 > g_pool.size()) return;
  Rec r;
  r.a = g_pool[g_pool_next++];
  r.b = g_pool[g_pool_next++];
  r.cat = cat;
  r.flops = flops;
  (void)hipEventRecord(r.a, stream);
  g_recs.push_back(r);
}

void osvos_prof_end(hipStream_t stream) {
  if (!g_on || g_recs.empty()) return;
  (void)hipEventRecord(g_recs.back().b, stream);
}

extern "C" {

int osvos_prof_start(int max_records) {
  OSVOS_ARG_CHECK(max_records > 0, "prof_start: max_records %d", max_records);
  const size_t had = g_pool.size();
  while (g_pool.size() < (size_t)2 * max_records) {
    hipEvent_t e;
    OSVOS_HIP_CHECK(hipEventCreate(&e));
    g_pool.push_back(e);
  }
  // a fresh hipEvent_t gets its backing signal at its FIRST record; done inside a timed region that costs tens of microseconds per
  // event (measured: a 20-step region lost 60 ms to ~2,700 first records).  Pay it here, outside.
  for (size_t i = had; i < g_pool.size(); ++i) OSVOS_HIP_CHECK(hipEventRecord(g_pool[i], nullptr));
  if (g_pool.size() > had) OSVOS_HIP_CHECK(hipStreamSynchronize(nullptr));
  g_recs.clear();
  g_recs.reserve(max_records);
  g_pool_next = 0;
  g_on = true;
  return 0;
}

// pause / resume without losing what was recorded: bench.py samples every n-th step of its timed region (an event pair around each of
// ~56 launches costs ~0.3 ms of a 5 ms step when left on for every step)
int osvos_prof_pause(int paused) {
  const int was = g_on ? 0 : 1;
  g_on = !paused && !g_pool.empty();
  return was;
}

// ms[c], flops[c], count[c] for c < OSVOS_PROF_NCAT; call after synchronising the stream
int osvos_prof_stop(double* ms, double* flops, long* count) {
  g_on = false;
  for (int c = 0; c < OSVOS_PROF_NCAT; ++c) { ms[c] = 0; flops[c] = 0; count[c] = 0; }
  for (const Rec& r : g_recs) {
    float t = 0.f;
    OSVOS_HIP_CHECK(hipEventSynchronize(r.b));
    OSVOS_HIP_CHECK(hipEventElapsedTime(&t, r.a, r.b));
    ms[r.cat] += t;
    flops[r.cat] += r.flops;
    count[r.cat] += 1;
  }
  g_recs.clear();
  g_pool_next = 0;
  return 0;
}

}  // extern "C"
