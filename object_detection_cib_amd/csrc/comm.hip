// Data-parallel collectives: RCCL called directly on the caller's HIP stream.
//
// Replaces what Lightning's strategy=ddp / sync_batchnorm=True set up for the reference
// (kod/configs/trainer/ddp.yaml:4-9): torch DDP's bucketed gradient all-reduce and SyncBatchNorm's per-layer
// statistic exchange.  The collectives are plain stream-ordered enqueues (no work objects, no watchdog thread,
// no host synchronisation), so a whole training step - kernels and all-reduces - is capturable as one hipGraph.
//
// RCCL is resolved at run time from the library the process already has loaded (PyTorch ships one) or from a
// path the host passes; nothing here links against it, and a missing library is a loud error, not a fallback.
#include <dlfcn.h>
#include <string.h>

#include "kodhip_common.h"

namespace {

struct UniqueId { char internal[128]; };           // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* Comm;                                // ncclComm_t
enum { kInt8 = 0, kUint8 = 1, kFloat32 = 7, kFloat64 = 8 };      // ncclDataType_t
enum { kSum = 0 };                                 // ncclRedOp_t

struct Api {
  void* handle = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
} g_api;

template <typename F>
bool bind(F& fn, const char* name) {
  fn = reinterpret_cast<F>(dlsym(g_api.handle, name));
  return fn != nullptr;
}

int load_api(const char* path) {
  if (g_api.handle) return KOD_OK;
  const char* names[] = {path, "librccl.so", "librccl.so.1"};
  // a copy that is already mapped wins (one RCCL per process), then the given path, then the loader's search path
  for (int pass = 0; pass < 2 && !g_api.handle; ++pass)
    for (const char* n : names) {
      if (!n || !*n) continue;
      g_api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (g_api.handle) break;
    }
  KOD_CHECK_ARG(g_api.handle, "comm: cannot load RCCL (%s)", dlerror());
  bool ok = bind(g_api.GetUniqueId, "ncclGetUniqueId") && bind(g_api.CommInitRank, "ncclCommInitRank") &&
            bind(g_api.CommDestroy, "ncclCommDestroy") && bind(g_api.AllReduce, "ncclAllReduce") &&
            bind(g_api.Broadcast, "ncclBroadcast") && bind(g_api.GroupStart, "ncclGroupStart") &&
            bind(g_api.GroupEnd, "ncclGroupEnd") && bind(g_api.GetErrorString, "ncclGetErrorString");
  if (!ok) {
    g_api = Api();
    kodhip_set_error("comm: the RCCL library lacks a required entry point");
    return KOD_EARG;
  }
  return KOD_OK;
}

#define KOD_RCCL(call, what)                                                       \
  do {                                                                             \
    int r__ = (call);                                                              \
    if (r__ != 0) {                                                                \
      kodhip_set_error("%s: RCCL error %d (%s)", what, r__, g_api.GetErrorString(r__)); \
      return 1000 + r__;                                                           \
    }                                                                              \
  } while (0)

}  // namespace

extern "C" {

// rccl_path may be NULL / "" (use the copy already loaded into the process, else the loader's search path)
int kodhip_comm_load(const char* rccl_path) { return load_api(rccl_path); }

// rank 0 creates the 128-byte rendezvous id and hands it to the other ranks out of band
int kodhip_comm_unique_id(void* id128) {
  KOD_CHECK_ARG(id128, "comm_unique_id: null");
  if (int e = load_api(nullptr)) return e;
  KOD_RCCL(g_api.GetUniqueId(reinterpret_cast<UniqueId*>(id128)), "comm_unique_id");
  return KOD_OK;
}

// collective over all ranks: creates this rank's communicator on the current HIP device
int kodhip_comm_init(void** comm, const void* id128, int rank, int world) {
  KOD_CHECK_ARG(comm && id128 && world >= 1 && rank >= 0 && rank < world, "comm_init: bad args (rank %d of %d)", rank, world);
  if (int e = load_api(nullptr)) return e;
  UniqueId id;
  memcpy(&id, id128, sizeof(id));
  Comm c = nullptr;
  KOD_RCCL(g_api.CommInitRank(&c, world, id, rank), "comm_init");
  *comm = c;
  return KOD_OK;
}

int kodhip_comm_destroy(void* comm) {
  if (!comm) return KOD_OK;
  KOD_CHECK_ARG(g_api.handle, "comm_destroy: RCCL not loaded");
  KOD_RCCL(g_api.CommDestroy(comm), "comm_destroy");
  return KOD_OK;
}

// in-place sum over ranks of `count` values: elem_bytes 4 = fp32 (gradient buckets), 8 = fp64 (SyncBN statistic sums)
int kodhip_comm_allreduce_sum(void* comm, void* buf, long count, int elem_bytes, hipStream_t stream) {
  KOD_CHECK_ARG(comm && buf && count > 0 && (elem_bytes == 4 || elem_bytes == 8), "comm_allreduce_sum: bad args");
  KOD_RCCL(g_api.AllReduce(buf, buf, (size_t)count, elem_bytes == 4 ? kFloat32 : kFloat64, kSum, comm, stream),
           "comm_allreduce_sum");
  return KOD_OK;
}

// out-of-place form: recv = sum over ranks of send (send is left untouched; SyncBN backward keeps the local sums)
int kodhip_comm_allreduce_sum_to(void* comm, const void* send, void* recv, long count, int elem_bytes, hipStream_t stream) {
  KOD_CHECK_ARG(comm && send && recv && count > 0 && (elem_bytes == 4 || elem_bytes == 8), "comm_allreduce_sum_to: bad args");
  KOD_RCCL(g_api.AllReduce(send, recv, (size_t)count, elem_bytes == 4 ? kFloat32 : kFloat64, kSum, comm, stream),
           "comm_allreduce_sum_to");
  return KOD_OK;
}

// ncclGroupStart / ncclGroupEnd: the collectives enqueued in between are launched as one fused operation - the
// SyncBN statistic exchanges of sibling layers (a CSP layer's main and short convs) travel together
int kodhip_comm_group_start(void) {
  KOD_CHECK_ARG(g_api.handle, "comm_group_start: RCCL not loaded");
  KOD_RCCL(g_api.GroupStart(), "comm_group_start");
  return KOD_OK;
}
int kodhip_comm_group_end(void) {
  KOD_CHECK_ARG(g_api.handle, "comm_group_end: RCCL not loaded");
  KOD_RCCL(g_api.GroupEnd(), "comm_group_end");
  return KOD_OK;
}

// in-place broadcast of `bytes` bytes from `root` (initial parameters and BatchNorm buffers)
int kodhip_comm_broadcast(void* comm, void* buf, long bytes, int root, hipStream_t stream) {
  KOD_CHECK_ARG(comm && buf && bytes > 0 && root >= 0, "comm_broadcast: bad args");
  KOD_RCCL(g_api.Broadcast(buf, buf, (size_t)bytes, kUint8, root, comm, stream), "comm_broadcast");
  return KOD_OK;
}

}  // extern "C"
