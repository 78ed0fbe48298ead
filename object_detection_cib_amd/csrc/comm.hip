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

// ---- SyncBN exchange over peer buffers ---------------------------------------------------------------------------
// sync_batchnorm (kod/configs/trainer/ddp.yaml:9) needs two fp64 sums per channel from every rank, 114 times per
// step, each one on the critical chain: as RCCL all-reduces that is 98-114 latency-bound collective launches.  On one
// node every GPU can map every other GPU's memory (xGMI), so each rank owns a small exchange buffer that all ranks
// open through HIP IPC; the BatchNorm finalize / coefficient kernels publish their sums there and read the other
// ranks' directly (csrc/bn_act.hip: peer_allreduce2) - one hop, no collective launch, no communicator ordering (the
// forward CSP branches stay on their side streams).  The buffer is the one device allocation this library owns
// (fine-grained, so that system-scope stores / loads of other agents bypass the caches); kodhip_peer_destroy frees it.
namespace {

struct PeerComm {
  int rank = 0, world = 1;
  long granules = 0;
  unsigned char* local = nullptr;               // [256 B header: seq, timeout flag][granules * 8 B]
  int* host_flag = nullptr;                     // pinned host copy of the verdict (read without synchronising the device)
  void* mapped[KOD_PEER_MAX] = {};
  KodPeerView view = {};
};
constexpr long PEER_HEADER = 256;

// bumps this rank's step number (published in the buffer header) and compares it with every peer's: when this rank
// begins step s a peer is at s - 1 (still finishing: its last exchanges needed this rank's step s - 1 publishes) or
// already at s - anything else means the ranks' step counters diverged (one extra training forward on one rank, uneven
// batch counts) and the run is condemned at once instead of after a minute of polling.
__global__ void peer_step_begin_kernel(unsigned int* seq, KodPeerView pv) {
  const int r = threadIdx.x;
  unsigned int mine = 0;
  if (r == 0) { mine = *seq + 1u; __hip_atomic_store(seq, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
  mine = __shfl(mine, 0, 64);
  if (r < pv.world && r != pv.rank) {
    const unsigned int* ps = (const unsigned int*)((const unsigned char*)pv.peers[r] - 256);
    const unsigned int theirs = __hip_atomic_load(ps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const int d = (int)(theirs - mine);
    if (d > 0 || d < -1) {
      __hip_atomic_store(pv.timeout_flag, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(pv.host_flag, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// transport self-test / generic form: out[i] = sum over ranks of in[i] (pairs of values per wave, like the BN kernels)
__device__ __forceinline__ void peer_allreduce_f64_body(const double* in, double* out, int n, const KodPeerView& pv, unsigned int slot,
                                                        int block) {
  const int pair = block * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int i0 = 2 * pair, i1 = 2 * pair + 1;
  if (i0 >= n) return;
  double s0 = in[i0], s1 = i1 < n ? in[i1] : 0.0;
  const unsigned int seq = *pv.seq;
  const int g = lane & 3;
  const unsigned long long gi = (unsigned long long)slot + 2ull * (unsigned long long)(g < 2 ? i0 : (i1 < n ? i1 : i0)) + (g & 1);
  const unsigned long long b0 = (unsigned long long)__double_as_longlong(s0), b1 = (unsigned long long)__double_as_longlong(s1);
  const bool second_ok = i1 < n;
  if (lane < 4 && (g < 2 || second_ok)) {
    const unsigned long long bits = g < 2 ? b0 : b1;
    __hip_atomic_store(pv.peers[pv.rank] + gi, ((unsigned long long)seq << 32) | (unsigned int)((g & 1) ? (bits >> 32) : bits),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  unsigned int got = 0;
  bool bad = false;
  const int r = lane >> 2;
  if (r < pv.world && (g < 2 || second_ok)) got = (unsigned int)kod_peer_poll(pv, pv.peers[r] + gi, seq, bad);
  double t0 = 0.0, t1 = 0.0;
  for (int q = 0; q < pv.world; ++q) {
    const unsigned long long lo0 = __shfl(got, 4 * q + 0, 64), hi0 = __shfl(got, 4 * q + 1, 64);
    const unsigned long long lo1 = __shfl(got, 4 * q + 2, 64), hi1 = __shfl(got, 4 * q + 3, 64);
    t0 += __longlong_as_double((long long)((hi0 << 32) | lo0));
    t1 += __longlong_as_double((long long)((hi1 << 32) | lo1));
  }
  if (__ballot(bad) != 0ull) t0 = t1 = __longlong_as_double(0x7ff8000000000000ll);
  if (lane == 0) {
    out[i0] = t0;
    if (second_ok) out[i1] = t1;
  }
}

__global__ void peer_allreduce_f64_kernel(const double* in, double* out, int n, KodPeerView pv, unsigned int slot) {
  peer_allreduce_f64_body(in, out, n, pv, slot, blockIdx.x);
}

// every rank of a one-process exchange in ONE dispatch (blockIdx.y = rank): the ranks' blocks are co-resident by
// construction, whatever hardware queue a stream would have been given (eight polling launches on eight streams share
// four queues by default: a launch queued behind the one that waits for it never starts)
struct PeerMultiArgs {
  KodPeerView pv[KOD_PEER_MAX];
  const double* in[KOD_PEER_MAX];
  double* out[KOD_PEER_MAX];
};
__global__ void peer_allreduce_f64_multi_kernel(PeerMultiArgs a, int n, unsigned int slot) {
  const int r = blockIdx.y;
  peer_allreduce_f64_body(a.in[r], a.out[r], n, a.pv[r], slot, blockIdx.x);
}

#define KOD_HIP(call, what)                                                   \
  do {                                                                        \
    hipError_t e__ = (call);                                                  \
    if (e__ != hipSuccess) {                                                  \
      kodhip_set_error("%s: %s", what, hipGetErrorString(e__));               \
      return (int)e__;                                                        \
    }                                                                         \
  } while (0)

}  // namespace

extern "C" {

// this rank's exchange buffer: `granules` 8-byte granules (4 per channel and exchange site), zero-initialised
int kodhip_peer_create(void** peer, int rank, int world, long granules) {
  KOD_CHECK_ARG(peer && world >= 1 && world <= KOD_PEER_MAX && rank >= 0 && rank < world && granules > 0,
                "peer_create: bad args (rank %d of %d, at most %d ranks of one node)", rank, world, KOD_PEER_MAX);
  PeerComm* c = new PeerComm();
  c->rank = rank; c->world = world; c->granules = granules;
  const size_t bytes = PEER_HEADER + (size_t)granules * 8;
  void* p = nullptr;
  // fine-grained or nothing: relaxed system-scope polling across GPUs is only coherent on fine-grained memory, so a
  // failed allocation is an error the caller answers by keeping the RCCL exchanges (no coarse-grained fallback)
  hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    delete c;
    kodhip_set_error("peer_create: fine-grained device memory for the exchange buffer: %s", hipGetErrorString(e));
    return (int)e;
  }
  c->local = (unsigned char*)p;
  e = hipMemset(p, 0, bytes);
  void* hf = nullptr;
  if (e == hipSuccess) e = hipHostMalloc(&hf, 64, hipHostMallocMapped);
  if (e == hipSuccess) { memset(hf, 0, 64); c->host_flag = (int*)hf; e = hipDeviceSynchronize(); }
  if (e != hipSuccess) {
    (void)hipFree(p);
    if (hf) (void)hipHostFree(hf);
    delete c; kodhip_set_error("peer_create: %s", hipGetErrorString(e)); return (int)e;
  }
  *peer = c;
  return KOD_OK;
}

// 64-byte HIP IPC handle of the buffer, to hand to the other ranks out of band
int kodhip_peer_export(void* peer, void* handle64) {
  KOD_CHECK_ARG(peer && handle64, "peer_export: null");
  PeerComm* c = (PeerComm*)peer;
  hipIpcMemHandle_t h;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "HIP IPC handle size");
  KOD_HIP(hipIpcGetMemHandle(&h, c->local), "peer_export");
  memcpy(handle64, &h, 64);
  return KOD_OK;
}

// handles: world x 64 bytes in rank order (this rank's own entry is ignored: it uses its local pointer)
int kodhip_peer_connect(void* peer, const void* handles) {
  KOD_CHECK_ARG(peer && handles, "peer_connect: null");
  PeerComm* c = (PeerComm*)peer;
  for (int r = 0; r < c->world; ++r) {
    if (r == c->rank) { c->mapped[r] = c->local; continue; }
    hipIpcMemHandle_t h;
    memcpy(&h, (const char*)handles + 64 * r, 64);
    void* p = nullptr;
    KOD_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess), "peer_connect");
    c->mapped[r] = p;
  }
  c->view = KodPeerView{};
  for (int r = 0; r < c->world; ++r) c->view.peers[r] = (unsigned long long*)((unsigned char*)c->mapped[r] + PEER_HEADER);
  c->view.world = c->world; c->view.rank = c->rank;
  c->view.seq = (const unsigned int*)c->local;
  c->view.timeout_flag = (int*)(c->local + 64);
  void* hfd = nullptr;
  KOD_HIP(hipHostGetDevicePointer(&hfd, c->host_flag, 0), "peer_connect");
  c->view.host_flag = (int*)hfd;
  c->view.max_spins = 1l << 26;              // inside a training step: about a minute
  return KOD_OK;
}

// The same connection for ranks that live in ONE process (one host thread driving several GPUs with peer access, or
// several exchange buffers on one GPU): `peers` = the world's kodhip_peer_create handles in rank order, this rank's own
// entry included; the other ranks' buffers are used through their device pointers, no IPC handle involved.
int kodhip_peer_connect_local(void* peer, void* const* peers) {
  KOD_CHECK_ARG(peer && peers, "peer_connect_local: null");
  PeerComm* c = (PeerComm*)peer;
  for (int r = 0; r < c->world; ++r) {
    PeerComm* o = (PeerComm*)peers[r];
    KOD_CHECK_ARG(o && o->world == c->world && o->rank == r && o->granules == c->granules && o->local,
                  "peer_connect_local: entry %d is not rank %d of the same %d-rank exchange", r, r, c->world);
  }
  KOD_CHECK_ARG(peers[c->rank] == peer, "peer_connect_local: this rank's own entry must be its handle");
  c->view = KodPeerView{};
  for (int r = 0; r < c->world; ++r) {
    c->mapped[r] = nullptr;            // (nothing to close: the buffers belong to the other handles)
    c->view.peers[r] = (unsigned long long*)(((PeerComm*)peers[r])->local + PEER_HEADER);
  }
  c->view.world = c->world; c->view.rank = c->rank;
  c->view.seq = (const unsigned int*)c->local;
  c->view.timeout_flag = (int*)(c->local + 64);
  void* hfd = nullptr;
  KOD_HIP(hipHostGetDevicePointer(&hfd, c->host_flag, 0), "peer_connect_local");
  c->view.host_flag = (int*)hfd;
  c->view.max_spins = 1l << 26;
  return KOD_OK;
}

int kodhip_peer_view_bytes(void) { return (int)sizeof(KodPeerView); }

// copies the KodPeerView the BatchNorm kernels take by value (device pointers of every rank's granule area)
int kodhip_peer_view(void* peer, void* view_out) {
  KOD_CHECK_ARG(peer && view_out, "peer_view: null");
  PeerComm* c = (PeerComm*)peer;
  KOD_CHECK_ARG(c->view.world == c->world, "peer_view: call kodhip_peer_connect first");
  memcpy(view_out, &c->view, sizeof(KodPeerView));
  return KOD_OK;
}

// once per training step, before its first exchange (stream-ordered; part of a captured step)
int kodhip_peer_step_begin(void* peer, hipStream_t stream) {
  KOD_CHECK_ARG(peer, "peer_step_begin: null");
  PeerComm* c = (PeerComm*)peer;
  KOD_CHECK_ARG(c->view.world == c->world, "peer_step_begin: call kodhip_peer_connect first");
  hipLaunchKernelGGL(peer_step_begin_kernel, dim3(1), dim3(64), 0, stream, (unsigned int*)c->local, c->view);
  KOD_LAUNCH_CHECK("peer_step_begin");
  return KOD_OK;
}

// out[i] = sum over ranks of in[i] (fp64, n values, granules [slot, slot + 2 * n)): the transport alone - start-up
// self-test against an RCCL all-reduce, and the exchange of anything that is not a BatchNorm statistic
int kodhip_peer_allreduce_f64(void* peer, const double* in, double* out, int n, unsigned int slot, hipStream_t stream) {
  KOD_CHECK_ARG(peer && in && out && n > 0, "peer_allreduce_f64: bad args");
  PeerComm* c = (PeerComm*)peer;
  KOD_CHECK_ARG((long)slot + 2l * n <= c->granules, "peer_allreduce_f64: slot range beyond the exchange buffer");
  const int pairs = (n + 1) / 2;
  KodPeerView v = c->view;
  v.max_spins = 1l << 22;                    // the transport alone (self-test): a few seconds
  hipLaunchKernelGGL(peer_allreduce_f64_kernel, dim3(cdiv(pairs, 4)), dim3(256), 0, stream, in, out, n, v, slot);
  KOD_LAUNCH_CHECK("peer_allreduce_f64");
  return KOD_OK;
}

// The same exchange for ALL ranks of a one-process group (kodhip_peer_connect_local) as one dispatch: ins / outs = the
// ranks' n-value vectors in rank order.  The whole grid must be resident at once (the ranks' blocks wait for each other):
// ceil(n / 8) * world blocks of 256 threads, at most 2048.
int kodhip_peer_allreduce_f64_multi(void* const* peers, int world, const double* const* ins, double* const* outs, int n,
                                    unsigned int slot, hipStream_t stream) {
  KOD_CHECK_ARG(peers && ins && outs && world >= 1 && world <= KOD_PEER_MAX && n > 0, "peer_allreduce_f64_multi: bad args");
  PeerMultiArgs a = {};
  for (int r = 0; r < world; ++r) {
    PeerComm* c = (PeerComm*)peers[r];
    KOD_CHECK_ARG(c && c->world == world && c->rank == r && c->view.world == world && ins[r] && outs[r],
                  "peer_allreduce_f64_multi: entry %d is not a connected rank %d of %d", r, r, world);
    KOD_CHECK_ARG((long)slot + 2l * n <= c->granules, "peer_allreduce_f64_multi: slot range beyond the exchange buffer");
    a.pv[r] = c->view;
    a.pv[r].max_spins = 1l << 22;
    a.in[r] = ins[r]; a.out[r] = outs[r];
  }
  const int pairs = (n + 1) / 2;
  KOD_CHECK_ARG((long)cdiv(pairs, 4) * world <= 2048, "peer_allreduce_f64_multi: the grid must be resident at once (n too large)");
  hipLaunchKernelGGL(peer_allreduce_f64_multi_kernel, dim3(cdiv(pairs, 4), world), dim3(256), 0, stream, a, n, slot);
  KOD_LAUNCH_CHECK("peer_allreduce_f64_multi");
  return KOD_OK;
}

// non-zero when an exchange failed since the last call (1: a poll gave up, 2: the ranks' step counters diverged);
// synchronises the device and resets the flag
int kodhip_peer_timed_out(void* peer, int* flag) {
  KOD_CHECK_ARG(peer && flag, "peer_timed_out: null");
  PeerComm* c = (PeerComm*)peer;
  int v = 0;
  KOD_HIP(hipMemcpy(&v, c->local + 64, sizeof(int), hipMemcpyDeviceToHost), "peer_timed_out");
  if (v) { int z = 0; KOD_HIP(hipMemcpy(c->local + 64, &z, sizeof(int), hipMemcpyHostToDevice), "peer_timed_out"); }
  if (c->host_flag) *(volatile int*)c->host_flag = 0;
  *flag = v;
  return KOD_OK;
}

// the same verdict WITHOUT touching the device: the kernels mirror it into pinned host memory, so the training loop
// can look at it once per step (engine: before every step / replay) and raise instead of training on NaN statistics
int kodhip_peer_status(void* peer, int* flag) {
  KOD_CHECK_ARG(peer && flag, "peer_status: null");
  PeerComm* c = (PeerComm*)peer;
  *flag = c->host_flag ? *(volatile int*)c->host_flag : 0;
  return KOD_OK;
}

int kodhip_peer_destroy(void* peer) {
  if (!peer) return KOD_OK;
  PeerComm* c = (PeerComm*)peer;
  (void)hipDeviceSynchronize();
  for (int r = 0; r < c->world; ++r)
    if (r != c->rank && c->mapped[r]) (void)hipIpcCloseMemHandle(c->mapped[r]);
  if (c->local) (void)hipFree(c->local);
  if (c->host_flag) (void)hipHostFree(c->host_flag);
  delete c;
  return KOD_OK;
}

}  // extern "C"
