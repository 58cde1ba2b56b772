// RCCL through the C ABI (SURVEY.md 8b "osvos_comm_{init,allreduce,destroy}"): the ONE exchange step of the data-parallel loop -- the sum of
// the flat gradient buffer over the ranks, once per optimizer step (extends reference train_parent.py:163-172, which has no distributed
// code) -- for callers that do not go through torch.distributed, and for overlapping it with the backward: the chunked form waits, on a
// communication stream, for the gradient-ready events osvos_net_backward records (osvos_net_arm_grad_events) and reduces each gradient group
// as soon as it is complete.  No per-collective host bookkeeping: N x (hipStreamWaitEvent + ncclAllReduce).
//
// librccl is bound at run time (dlopen): the library has no link-time dependency on it, a process that already carries a copy (PyTorch
// bundles one) shares it, and single-GPU users never load it.
#include <dlfcn.h>
#include <string.h>

#include "common.h"

namespace {

typedef int ncclResult_t;                    // ncclSuccess = 0
typedef struct ncclComm* ncclComm_t;
struct UniqueId { char internal[128]; };     // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
constexpr int kNcclFloat32 = 7, kNcclFloat64 = 8, kNcclSum = 0;    // rccl.h: ncclFloat32 = 7, ncclFloat64 = 8, ncclSum = 0

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(UniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, UniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
  char why[256] = "";                        // the loader's message, captured right after the failing dlopen / dlsym
};

Rccl& rccl() {
  static Rccl r = [] {
    Rccl q;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
      q.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (q.lib) break;
      const char* e = dlerror();             // (dlerror() clears the state: read it once)
      snprintf(q.why, sizeof(q.why), "%s", e ? e : "dlopen failed");
    }
    if (!q.lib) return q;
    q.why[0] = 0;
    q.GetUniqueId = reinterpret_cast<decltype(q.GetUniqueId)>(dlsym(q.lib, "ncclGetUniqueId"));
    q.CommInitRank = reinterpret_cast<decltype(q.CommInitRank)>(dlsym(q.lib, "ncclCommInitRank"));
    q.CommDestroy = reinterpret_cast<decltype(q.CommDestroy)>(dlsym(q.lib, "ncclCommDestroy"));
    q.AllReduce = reinterpret_cast<decltype(q.AllReduce)>(dlsym(q.lib, "ncclAllReduce"));
    q.GroupStart = reinterpret_cast<decltype(q.GroupStart)>(dlsym(q.lib, "ncclGroupStart"));
    q.GroupEnd = reinterpret_cast<decltype(q.GroupEnd)>(dlsym(q.lib, "ncclGroupEnd"));
    q.GetErrorString = reinterpret_cast<decltype(q.GetErrorString)>(dlsym(q.lib, "ncclGetErrorString"));
    q.ok = q.GetUniqueId && q.CommInitRank && q.CommDestroy && q.AllReduce && q.GroupStart && q.GroupEnd && q.GetErrorString;
    if (!q.ok) {
      const char* e = dlerror();
      snprintf(q.why, sizeof(q.why), "%s", e ? e : "a required ncclXxx symbol is missing");
    }
    return q;
  }();
  return r;
}

#define OSVOS_NCCL_CHECK(expr)                                                                         \
  do {                                                                                                 \
    const ncclResult_t r_ = (expr);                                                                    \
    if (r_ != 0) {                                                                                     \
      osvos_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, rccl().GetErrorString(r_));         \
      return 1000 + (int)r_;                                                                           \
    }                                                                                                  \
  } while (0)

int need_rccl() {
  if (!rccl().ok) {
    osvos_set_error("librccl.so could not be loaded (dlopen / dlsym): %s", rccl().why[0] ? rccl().why : "symbols missing");
    return -2;
  }
  return 0;
}

}  // namespace

extern "C" {

int osvos_comm_unique_id(void* id128) {
  OSVOS_ARG_CHECK(id128 != nullptr, "comm_unique_id: null pointer");
  if (int rc = need_rccl()) return rc;
  UniqueId id;
  OSVOS_NCCL_CHECK(rccl().GetUniqueId(&id));
  memcpy(id128, id.internal, sizeof(id.internal));
  return 0;
}

int osvos_comm_init(void** comm, int rank, int world, const void* id128) {
  OSVOS_ARG_CHECK(comm && id128 && world >= 1 && rank >= 0 && rank < world, "comm_init: bad arguments (rank %d of %d)", rank, world);
  if (int rc = need_rccl()) return rc;
  UniqueId id;
  memcpy(id.internal, id128, sizeof(id.internal));
  ncclComm_t c = nullptr;
  OSVOS_NCCL_CHECK(rccl().CommInitRank(&c, world, id, rank));
  *comm = c;
  return 0;
}

int osvos_comm_allreduce_f32(void* comm, float* buf, size_t count, void* stream) {
  OSVOS_ARG_CHECK(comm && buf && count > 0, "comm_allreduce: bad arguments");
  if (int rc = need_rccl()) return rc;
  OSVOS_NCCL_CHECK(rccl().AllReduce(buf, buf, count, kNcclFloat32, kNcclSum, (ncclComm_t)comm, (hipStream_t)stream));
  return 0;
}

// float64 sum: class counts and loss statistics (exact for integers below 2^53; the fp32 form rounds counts above 2^24 -- about batch 41 at
// 854x480 -- and epoch loss sums near 1e7)
int osvos_comm_allreduce_f64(void* comm, double* buf, size_t count, void* stream) {
  OSVOS_ARG_CHECK(comm && buf && count > 0, "comm_allreduce_f64: bad arguments");
  if (int rc = need_rccl()) return rc;
  OSVOS_NCCL_CHECK(rccl().AllReduce(buf, buf, count, kNcclFloat64, kNcclSum, (ncclComm_t)comm, (hipStream_t)stream));
  return 0;
}

int osvos_comm_allreduce_chunks_f32(void* comm, float* buf, const size_t* first, const size_t* count, void* const* ready_events, int n,
                                    void* comm_stream) {
  OSVOS_ARG_CHECK(comm && buf && first && count && n > 0 && n <= 64, "comm_allreduce_chunks: bad arguments");
  if (int rc = need_rccl()) return rc;
  hipStream_t st = (hipStream_t)comm_stream;
  for (int k = 0; k < n; ++k) {
    if (ready_events != nullptr && ready_events[k] != nullptr) OSVOS_HIP_CHECK(hipStreamWaitEvent(st, (hipEvent_t)ready_events[k], 0));
    if (count[k] == 0) continue;
    OSVOS_NCCL_CHECK(rccl().AllReduce(buf + first[k], buf + first[k], count[k], kNcclFloat32, kNcclSum, (ncclComm_t)comm, st));
  }
  return 0;
}

int osvos_comm_destroy(void* comm) {
  if (comm == nullptr) return 0;
  if (int rc = need_rccl()) return rc;
  OSVOS_NCCL_CHECK(rccl().CommDestroy((ncclComm_t)comm));
  return 0;
}

}  // extern "C"
