// Shared device/host helpers for libkodhip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define KOD_OK 0
#define KOD_EARG (-1)

extern "C" void kodhip_set_error(const char* fmt, ...);

#define KOD_CHECK_ARG(cond, ...)                 \
  do {                                           \
    if (!(cond)) {                               \
      kodhip_set_error(__VA_ARGS__);             \
      return KOD_EARG;                           \
    }                                            \
  } while (0)

#define KOD_LAUNCH_CHECK(name)                                             \
  do {                                                                     \
    hipError_t e__ = hipGetLastError();                                    \
    if (e__ != hipSuccess) {                                               \
      kodhip_set_error("%s: %s", name, hipGetErrorString(e__));            \
      return (int)e__;                                                     \
    }                                                                      \
  } while (0)

// A 16-byte load of data this launch reads exactly once and nothing reads again soon (the pre-BN tensor behind its apply
// pass, dA behind the backward apply, weight-gradient slabs behind their reduction, gradients behind the optimizer, the
// fp32 input image): non-temporal, so that it does not push the tensors the NEXT launches re-read out of L2 / Infinity Cache.
// Measured on the replayed step: + 0.7 % for the two apply passes alone (LOG round 5); same bits.
#ifndef KOD_NO_NT
template <typename V> __device__ __forceinline__ V kod_load_once(const void* p) { return __builtin_nontemporal_load(reinterpret_cast<const V*>(p)); }
#else
template <typename V> __device__ __forceinline__ V kod_load_once(const void* p) { return *reinterpret_cast<const V*>(p); }
#endif

__device__ __forceinline__ float bf2f(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t f2bf(float v) { return (bf16_t)v; }

__device__ __forceinline__ float silu_f(float z) { return z / (1.0f + __expf(-z)); }
__device__ __forceinline__ float sigmoid_f(float z) { return 1.0f / (1.0f + __expf(-z)); }

// ---- BatchNorm + SiLU elementwise arithmetic shared by the passes that form it (bn_act.hip, the data gradients' fused
// reduction epilogue in conv_igemm.hip, the fused stem backward in conv_wgrad.hip).  These passes cost VALU issue slots
// that the co-running weight gradients want, so: explicit FMAs (the library is built with -ffp-contract=off; 21.8 -> 15 issue
// cycles per element in the backward pass) and sigmoid = rcp(1 + exp2(-log2e * z)): v_exp_f32 + v_rcp_f32, 1 ulp each, no
// Newton step (the results are rounded to bf16).  (How much is left in them: a backward apply with no sigmoid at all - dz taken
// from the data gradient's epilogue - is 6 % shorter in isolation and moves the step by 0.3 %, LOG round 5; round 2's
// "+ 6 % step with the sigmoid replaced by the identity" came from a diverged run on a chip that then clocks higher.)  (Folding -log2e into per-channel constants would save the multiply but costs sixteen
// registers: an occupancy step in the apply pass, spills in the 128-register conv tiles.)
#define KOD_NEG_LOG2E (-1.4426950408889634f)
__device__ __forceinline__ float kod_sigmoid_l2(float zl) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(zl));
}
// g * silu'(z) with sg = sigmoid(z):  g * sg * (1 + z * (1 - sg))
__device__ __forceinline__ float kod_silu_bwd(float g, float z, float sg) {
  const float t = __builtin_fmaf(-z, sg, z);
  return g * __builtin_fmaf(sg, t, sg);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- SyncBN peer exchange (csrc/comm.hip: kodhip_peer_*): every rank of the node maps every other rank's exchange
// buffer; a value travels as two 8-byte {tag : 32 | payload : 32} granules written by ONE system-scope store each (the
// tag is the step's sequence number: a reader polls until it sees it - no separate flag, no ordering between stores).
#define KOD_PEER_MAX 8
struct KodPeerView {
  unsigned long long* peers[KOD_PEER_MAX];   // device-visible base of rank r's granule area (own rank: the local buffer)
  int world, rank;
  const unsigned int* seq;                   // device: sequence number of the current step (kodhip_peer_step_begin)
  int* timeout_flag;                         // device: 1 when a poll gave up (a peer never published), 2 when a peer's tag or
                                             // step number ran AHEAD of this rank's (the ranks' step counters diverged)
  int* host_flag;                            // the same verdict in pinned host memory: the training loop reads it without a
                                             // device synchronisation (kodhip_peer_status)
  long max_spins;                            // polls before giving up: ~a minute inside a training step (ranks may reach their
                                             // first exchange seconds apart), seconds in the start-up self-test
};

// One granule poll of the exchange: returns the granule once its tag is this step's.  A poll that gives up, a tag
// from the future (no rank can be a step ahead at a slot this rank has not read yet: its later exchanges of the step
// need this rank's later publishes) or a verdict already raised earlier in the run (every later exchange then leaves at
// once instead of spinning for a minute each) sets `bad`: the caller turns the sums into NaN, so that the statistics
// are never built from a stale payload and the loss trips.
__device__ __forceinline__ unsigned long long kod_peer_poll(const KodPeerView& pv, const unsigned long long* src,
                                                            unsigned int seq, bool& bad) {
  const int raised = __hip_atomic_load(pv.timeout_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned long long v = 0;
  long spins = 0;
  for (;;) {
    v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned int tag = (unsigned int)(v >> 32);
    if (tag == seq) break;
    int why = 0;
    if ((int)(tag - seq) > 0) why = 2;                 // desynchronised step counters
    else if (raised) why = raised;                     // the run is already condemned
    else if (++spins > pv.max_spins) why = 1;          // a peer is gone
    if (why) {
      bad = true;
      if (!raised) {
        __hip_atomic_store(pv.timeout_flag, why, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pv.host_flag, why, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      break;
    }
    __builtin_amdgcn_s_sleep(8);
  }
  return v;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline uint32_t magic_u32(uint32_t d) { return (uint32_t)(((1ull << 32) + d - 1) / d); }
