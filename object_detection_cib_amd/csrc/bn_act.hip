// Train-mode BatchNorm2d(eps=1e-3, momentum=.03) + SiLU around the conv kernels
// (kod/nn/networks/yolov5.py:24 Yolov5BatchNorm2d, kod/nn/layers/activations.py:7 SiLUInplace; the
// aten ops native_batch_norm / silu / their backwards and the CSPBlock residual add, csp.py:55-56).
//
// All tensors are channels-last bf16; per-channel statistics are fp32/fp64.  HBM-bound elementwise
// kernels: 16-byte (8-channel) accesses, each thread keeps a fixed channel chunk so the per-channel
// constants live in registers; reductions are wavefront shuffles -> LDS -> fixed-order partial slabs
// (deterministic, no float atomics).
#include "kodhip_common.h"
#include <stdlib.h>

namespace {

// Sum of T fp32 partials in fp64 by one wave, fixed order: 16-byte loads, four independent accumulation chains so
// that the loads of a pass are in flight together (these kernels are pure latency: one wave per channel).
__device__ __forceinline__ double wave_sum_partials(const float* p, int T, int lane) {
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  const int T4 = ((reinterpret_cast<uintptr_t>(p) & 15) == 0) ? (T & ~3) : 0;
#pragma unroll 4
  for (int t = lane * 4; t < T4; t += 256) {
    f32x4 v = *reinterpret_cast<const f32x4*>(p + t);
    a0 += (double)v[0]; a1 += (double)v[1]; a2 += (double)v[2]; a3 += (double)v[3];
  }
  for (int t = T4 + lane; t < T; t += 64) a0 += (double)p[t];
  return wave_sum_d((a0 + a1) + (a2 + a3));
}

// Two such sums (a channel's two moments) with ALL loads issued before the first add: called one after the other, the
// second sum's loads would start behind the first sum's shuffle reduction - a second memory round trip in a kernel that
// is nothing but latency.  Same order of additions as wave_sum_partials (bit-identical results).
__device__ __forceinline__ void wave_sum_partials2(const float* p, const float* q, int T, int lane, double& sp, double& sq) {
  const bool al = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(q)) & 15) == 0;
  if (!al || T > 4 * 256) {               // unaligned rows / more than the four unrolled passes (1024 slots): the plain form
    sp = wave_sum_partials(p, T, lane);
    sq = wave_sum_partials(q, T, lane);
    return;
  }
  const int T4 = T & ~3;
  const int t0 = lane * 4, t1 = t0 + 256 * 1, t2 = t0 + 256 * 2, t3 = t0 + 256 * 3;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4 pv[4], qv[4];
  const int ts[4] = {t0, t1, t2, t3};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    pv[i] = ts[i] < T4 ? *reinterpret_cast<const f32x4*>(p + ts[i]) : z;
    qv[i] = ts[i] < T4 ? *reinterpret_cast<const f32x4*>(q + ts[i]) : z;
  }
  float pt = 0.f, qt = 0.f;
  const bool tail = T4 + lane < T;
  if (tail) { pt = p[T4 + lane]; qt = q[T4 + lane]; }
  double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (ts[i] < T4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { a[e] += (double)pv[i][e]; b[e] += (double)qv[i][e]; }
    }
  }
  if (tail) { a[0] += (double)pt; b[0] += (double)qt; }
  sp = wave_sum_d((a[0] + a[1]) + (a[2] + a[3]));
  sq = wave_sum_d((b[0] + b[1]) + (b[2] + b[3]));
}

// ---------------------------------------------------------------- SyncBN over peer buffers
// One wave per channel holds the rank's two fp64 sums (s0, s1) in lane 0: publish them as four granules in the rank's
// own exchange buffer, then poll the same four granules of EVERY rank (lane = 4 * rank + granule) and add the ranks'
// values in rank order - every rank gets bit-identical totals.  Publishing precedes polling in every wave and waves
// do not depend on each other, so ranks can arrive in any order; a poll that never sees its tag gives up after
// max_spins polls (about a minute), raises the flag (device + pinned host copy) and turns the sums into NaN instead of
// hanging the GPU or folding the previous step's payload into the statistics (kod_peer_poll, kodhip_common.h).
__device__ __forceinline__ void peer_allreduce2(const KodPeerView& pv, unsigned int slot, int idx0, int idx1, int lane,
                                                double& s0, double& s1) {
  const unsigned int seq = *pv.seq;
  // lane 0 holds the values; lanes 0..3 store one granule each
  const unsigned long long b0 = __shfl((unsigned long long)__double_as_longlong(s0), 0, 64);
  const unsigned long long b1 = __shfl((unsigned long long)__double_as_longlong(s1), 0, 64);
  const int g = lane & 3;                                // 0: s0.lo, 1: s0.hi, 2: s1.lo, 3: s1.hi
  const unsigned long long gi = (unsigned long long)slot + 2ull * (unsigned long long)(g < 2 ? idx0 : idx1) + (g & 1);
  if (lane < 4) {
    const unsigned long long bits = g < 2 ? b0 : b1;
    const unsigned int payload = (unsigned int)((g & 1) ? (bits >> 32) : bits);
    __hip_atomic_store(pv.peers[pv.rank] + gi, ((unsigned long long)seq << 32) | payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  unsigned int got = 0;
  bool bad = false;
  const int r = lane >> 2;
  if (r < pv.world) got = (unsigned int)kod_peer_poll(pv, pv.peers[r] + gi, seq, bad);
  double t0 = 0.0, t1 = 0.0;
  for (int q = 0; q < pv.world; ++q) {                   // fixed rank order
    const unsigned long long lo0 = __shfl(got, 4 * q + 0, 64), hi0 = __shfl(got, 4 * q + 1, 64);
    const unsigned long long lo1 = __shfl(got, 4 * q + 2, 64), hi1 = __shfl(got, 4 * q + 3, 64);
    t0 += __longlong_as_double((long long)((hi0 << 32) | lo0));
    t1 += __longlong_as_double((long long)((hi1 << 32) | lo1));
  }
  if (__ballot(bad) != 0ull) t0 = t1 = __longlong_as_double(0x7ff8000000000000ll);     // never a stale payload in the sums
  s0 = t0; s1 = t1;
}

// ---------------------------------------------------------------- partial slabs -> fp64 sums
// in: part[2][C][T] fp32 ; out: sums[2][C] fp64.  One wave per (stat, channel).
__global__ void bn_reduce_partials_kernel(const float* part, double* sums, int C, int T) {
  int idx = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (idx >= 2 * C) return;
  int lane = threadIdx.x & 63;
  double s = wave_sum_partials(part + (size_t)idx * T, T, lane);
  if (lane == 0) sums[idx] = s;
}

// forward finalize: sums (already all-reduced over ranks when SyncBN) -> affine constants + running stats
__global__ void bn_finalize_kernel(const double* sums, double count, const float* gamma, const float* beta,
                                   float* running_mean, float* running_var, float momentum, float eps,
                                   float* scale, float* shift, float* mean_out, float* rstd_out, int C,
                                   int update_running) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double mean = sums[c] / count;
  double var = sums[C + c] / count - mean * mean;
  if (var < 0.0) var = 0.0;
  float rstd = (float)(1.0 / sqrt(var + (double)eps));
  float g = gamma[c], b = beta[c];
  float sc = g * rstd;
  scale[c] = sc;
  shift[c] = b - (float)mean * sc;
  mean_out[c] = (float)mean;
  rstd_out[c] = rstd;
  if (update_running) {
    double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// single-GPU fast path: partial slabs -> constants in ONE launch (one wave per channel).  PEER: SyncBN - the rank's
// sums are exchanged through the peer buffers inside the same launch (count = pixels of ALL ranks)
template <bool PEER>
__global__ void bn_finalize_fused_kernel(const float* part, int T, double count, const float* gamma,
                                         const float* beta, float* running_mean, float* running_var,
                                         float momentum, float eps, float* scale, float* shift, float* mean_out,
                                         float* rstd_out, int C, int update_running, KodPeerView pv, unsigned int slot) {
  int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (c >= C) return;
  int lane = threadIdx.x & 63;
  const float* p0 = part + (size_t)c * T;
  const float* p1 = part + (size_t)(C + c) * T;
  // the per-channel parameters are loaded up front: behind the reduction they would be a second, dependent memory round
  // trip in a kernel that is nothing but latency (one wave per channel, 57 launches on the critical chain)
  const float g = gamma[c], b = beta[c];
  const float rm0 = update_running ? running_mean[c] : 0.f, rv0 = update_running ? running_var[c] : 0.f;
  double s0, s1;
  wave_sum_partials2(p0, p1, T, lane, s0, s1);
  if constexpr (PEER) peer_allreduce2(pv, slot, c, C + c, lane, s0, s1);
  if (lane != 0) return;
  double mean = s0 / count;
  double var = s1 / count - mean * mean;
  if (var < 0.0) var = 0.0;
  float rstd = (float)(1.0 / sqrt(var + (double)eps));
  float sc = g * rstd;
  scale[c] = sc;
  shift[c] = b - (float)mean * sc;
  mean_out[c] = (float)mean;
  rstd_out[c] = rstd;
  if (update_running) {
    double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (1.f - momentum) * rm0 + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * rv0 + momentum * (float)unbiased;
  }
}

// Two units whose convolution ran as ONE launch (a CSP layer's main_conv + short_conv as N = 2 * Ch columns,
// kod/nn/layers/csp.py:87-88): the statistic slots are [2][2 * Ch][T]; blockIdx.y picks the unit (channels
// [half * Ch, (half + 1) * Ch) of both moments), each with its own parameters, running statistics and constants.
struct FinJob {
  const float* gamma; const float* beta; float* running_mean; float* running_var;
  float* scale; float* shift; float* mean; float* rstd;
  unsigned int slot;
};
template <bool PEER>
__global__ void bn_finalize_pair_kernel(const float* part, int T, double count, float momentum, float eps, int Ch,
                                        int update_running, FinJob j0, FinJob j1, KodPeerView pv) {
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (c >= Ch) return;
  const int half = blockIdx.y;
  const FinJob j = half ? j1 : j0;
  const int lane = threadIdx.x & 63;
  const float* p0 = part + (size_t)(half * Ch + c) * T;
  const float* p1 = part + (size_t)(2 * Ch + half * Ch + c) * T;
  const float g = j.gamma[c], b = j.beta[c];
  const float rm0 = update_running ? j.running_mean[c] : 0.f, rv0 = update_running ? j.running_var[c] : 0.f;
  double s0, s1;
  wave_sum_partials2(p0, p1, T, lane, s0, s1);
  if constexpr (PEER) peer_allreduce2(pv, j.slot, c, Ch + c, lane, s0, s1);
  if (lane != 0) return;
  double mean = s0 / count;
  double var = s1 / count - mean * mean;
  if (var < 0.0) var = 0.0;
  float rstd = (float)(1.0 / sqrt(var + (double)eps));
  float sc = g * rstd;
  j.scale[c] = sc;
  j.shift[c] = b - (float)mean * sc;
  j.mean[c] = (float)mean;
  j.rstd[c] = rstd;
  if (update_running) {
    double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    j.running_mean[c] = (1.f - momentum) * rm0 + momentum * (float)mean;
    j.running_var[c] = (1.f - momentum) * rv0 + momentum * (float)unbiased;
  }
}

// raw_moment: the second partial is sum dz*y (produced by the data-gradient epilogue, conv_igemm.hip MODE_PLAIN_BN)
// instead of sum dz*xhat; xhat = (y - mean)*rstd  =>  sum dz*xhat = rstd * (sum dz*y - mean * sum dz), in fp64.
template <bool PEER = false>
__device__ __forceinline__ void bn_bwd_coeffs_channel(const float* part, int T, double count, const float* gamma,
                                                      const float* mean, const float* rstd, float* dgamma, float* dbeta,
                                                      float* coef, int C, int raw_moment, int c, int lane,
                                                      const KodPeerView* pv = nullptr, unsigned int slot = 0) {
  const float* p0 = part + (size_t)c * T;
  const float* p1 = part + (size_t)(C + c) * T;
  const float g_ = gamma[c], rs_ = rstd[c], mu_ = mean[c];      // up front: not a second round trip behind the reduction
  double s0, s1;
  wave_sum_partials2(p0, p1, T, lane, s0, s1);
  if (raw_moment) s1 = (double)rs_ * (s1 - (double)mu_ * s0);      // (linear in the sums: ranks may add converted values)
  if (lane == 0) {          // parameter gradients keep the rank's own sums (the gradient all-reduce adds the ranks later)
    dbeta[c] = (float)s0;
    dgamma[c] = (float)s1;
  }
  if constexpr (PEER) peer_allreduce2(*pv, slot, c, C + c, lane, s0, s1);   // dX uses the sums over ALL ranks' pixels
  if (lane != 0) return;
  double g = g_, rs = rs_, mu = mu_;
  double S0 = s0 / count, S1 = s1 / count;
  coef[c] = (float)(g * rs);
  coef[C + c] = (float)(-g * rs * rs * S1);
  coef[2 * C + c] = (float)(-g * rs * S0 + g * rs * rs * mu * S1);
}

__global__ void bn_bwd_coeffs_fused_kernel(const float* part, int T, double count, const float* gamma,
                                           const float* mean, const float* rstd, float* dgamma, float* dbeta,
                                           float* coef, int C, int raw_moment) {
  int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (c >= C) return;
  bn_bwd_coeffs_channel(part, T, count, gamma, mean, rstd, dgamma, dbeta, coef, C, raw_moment, c, threadIdx.x & 63);
}

__global__ void bn_bwd_coeffs_peer_kernel(const float* part, int T, double count, const float* gamma,
                                          const float* mean, const float* rstd, float* dgamma, float* dbeta,
                                          float* coef, int C, int raw_moment, KodPeerView pv, unsigned int slot) {
  int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (c >= C) return;
  bn_bwd_coeffs_channel<true>(part, T, count, gamma, mean, rstd, dgamma, dbeta, coef, C, raw_moment, c, threadIdx.x & 63, &pv, slot);
}

// the same for TWO units in one launch (blockIdx.y = unit): a CSP layer's short_conv and main_conv reach this point of
// the backward pass together, and on the critical chain a launch costs more than the arithmetic
struct CoefJob { const float* part; int T; double count; const float* gamma; const float* mean; const float* rstd;
                 float* dgamma; float* dbeta; float* coef; int C; int raw_moment; };
__global__ void bn_bwd_coeffs_fused2_kernel(CoefJob j0, CoefJob j1) {
  const CoefJob& j = blockIdx.y == 0 ? j0 : j1;
  int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (c >= j.C) return;
  bn_bwd_coeffs_channel(j.part, j.T, j.count, j.gamma, j.mean, j.rstd, j.dgamma, j.dbeta, j.coef, j.C, j.raw_moment, c,
                        threadIdx.x & 63);
}

// The elementwise passes are pure HBM streams: each thread keeps U rows (U x 16 B per operand) in flight before it
// touches the first one - with a single load per iteration the chip holds too few bytes in flight to cover the
// HBM latency (measured 3.5-4.5 TB/s; Little's law wants >= 12 MB outstanding for 8 TB/s).
constexpr int U = 4;          // (the reduce / pair kernels; the two apply passes take it as a template parameter)

// ---------------------------------------------------------------- forward apply
// out[m][ocoff + c] = silu(y[m][c]*scale[c] + shift[c]) (+ res[m][rcoff + c])
// Launch shape (template + launch arguments, chosen per tensor by apply_shape() below): U rows of one block-contiguous chunk
// in flight per thread (rows m0, m0 + rpb, ...: a block reads U * rpb * C * 2 contiguous bytes per operand), blocks of up to
// MAXT threads, and - LDSK - the per-channel constants staged through LDS once per block instead of read from global memory
// by every thread: with many short blocks the constant loads otherwise outnumber the payload's (4 + 2 vector memory
// instructions per row here, 10 + 3 in the backward pass) and the pass becomes address-unit-bound
// (tools/micro/stream_apply.hip, profiles/r06_stream_apply.txt).
template <int U, bool LDSK, int MAXT>
__global__ __launch_bounds__(MAXT) void bn_silu_apply_kernel(const bf16_t* y, int ldy, const float* scale, const float* shift,
                                     const bf16_t* res, int ldr, int rcoff,
                                     bf16_t* out, int ldo, int ocoff, long M, int C, int rows_per_block_iter) {
  extern __shared__ float kconst[];           // LDSK: [scale C | shift C]
  const int CC = C >> 3;
  const int cc = threadIdx.x % CC;
  const int rl = threadIdx.x / CC;
  float sc[8], sh[8];
  if (LDSK) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) { kconst[c] = scale[c]; kconst[C + c] = shift[c]; }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = kconst[cc * 8 + e]; sh[e] = kconst[C + cc * 8 + e]; }
  }
  if (rl >= rows_per_block_iter) return;
  if (!LDSK) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = scale[cc * 8 + e]; sh[e] = shift[cc * 8 + e]; }
  }
  const long step = (long)gridDim.x * rows_per_block_iter * U;
  for (long m0 = (long)blockIdx.x * rows_per_block_iter * U + rl; m0 < M; m0 += step) {
    bf16x8 v[U], r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long m = m0 + (long)u * rows_per_block_iter;
      if (m < M) {
        v[u] = kod_load_once<bf16x8>(y + m * ldy + cc * 8);
        if (res) r[u] = *reinterpret_cast<const bf16x8*>(res + m * ldr + rcoff + cc * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long m = m0 + (long)u * rows_per_block_iter;
      if (m >= M) break;
      bf16x8 o;
      if (res) {
        // the reference adds in fp32 on the fp32 activation; here the activation is rounded to bf16 only when it
        // is materialised, so add before the single rounding.
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float yv = (float)v[u][e];
          const float z = __builtin_fmaf(yv, sc[e], sh[e]);
          const float sg = kod_sigmoid_l2(KOD_NEG_LOG2E * z);
          o[e] = (bf16_t)__builtin_fmaf(z, sg, (float)r[u][e]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float yv = (float)v[u][e];
          const float z = __builtin_fmaf(yv, sc[e], sh[e]);
          o[e] = (bf16_t)(z * kod_sigmoid_l2(KOD_NEG_LOG2E * z));
        }
      }
#ifdef KOD_BN_NTST
      __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(out + m * ldo + ocoff + cc * 8));
#else
      *reinterpret_cast<bf16x8*>(out + m * ldo + ocoff + cc * 8) = o;
#endif
    }
  }
}

// Two units whose pre-BN outputs are the two channel halves of ONE tensor y[m][2 * Ch] (a CSP layer's main_conv +
// short_conv computed by one convolution launch): one pass over y, each half to its own destination slice.
__global__ __launch_bounds__(256) void bn_silu_apply_pair_kernel(const bf16_t* y, int ldy,
                                     const float* scale0, const float* shift0, bf16_t* out0, int ldo0, int ocoff0,
                                     const float* scale1, const float* shift1, bf16_t* out1, int ldo1, int ocoff1,
                                     long M, int Ch, int rows_per_block_iter) {
  const int CC = Ch >> 2;                     // 16-byte chunks of a row of y (2 * Ch channels)
  const int cc = threadIdx.x % CC;
  const int rl = threadIdx.x / CC;
  if (rl >= rows_per_block_iter) return;
  const bool second = cc >= (CC >> 1);
  const int c0 = (second ? cc - (CC >> 1) : cc) * 8;          // first channel of this chunk inside its unit
  const float* scale = second ? scale1 : scale0;
  const float* shift = second ? shift1 : shift0;
  bf16_t* out = second ? out1 + ocoff1 : out0 + ocoff0;
  const int ldo = second ? ldo1 : ldo0;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sc[e] = scale[c0 + e]; sh[e] = shift[c0 + e]; }
  const long stride = (long)gridDim.x * rows_per_block_iter;
  for (long m0 = (long)blockIdx.x * rows_per_block_iter + rl; m0 < M; m0 += stride * U) {
    bf16x8 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long m = m0 + u * stride;
      if (m < M) v[u] = *reinterpret_cast<const bf16x8*>(y + m * ldy + cc * 8);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long m = m0 + u * stride;
      if (m >= M) break;
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float yv = (float)v[u][e];
        const float z = __builtin_fmaf(yv, sc[e], sh[e]);
        o[e] = (bf16_t)(z * kod_sigmoid_l2(KOD_NEG_LOG2E * z));
      }
      *reinterpret_cast<bf16x8*>(out + m * ldo + c0) = o;
    }
  }
}

// ---------------------------------------------------------------- backward reduce
// dz = dA * silu'(z), z = y*scale + shift ; xhat = (y - mean)*rstd
// part[0][c][blk] = sum dz ; part[1][c][blk] = sum dz*xhat
__global__ __launch_bounds__(256) void bn_silu_bwd_reduce_kernel(const bf16_t* dA, int lda, int dacoff, const bf16_t* y, int ldy,
                                          const float* scale, const float* shift, const float* mean,
                                          const float* rstd, float* part, long M, int C, int rpb) {
  extern __shared__ float sm[];   // [rpb][CC][16]
  const int CC = C >> 3;
  const int cc = threadIdx.x % CC;
  const int rl = threadIdx.x / CC;
  float s0[8], s1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s0[e] = s1[e] = 0.f;
  if (rl < rpb) {
    float sc[8], sh[8], mu[8], rs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sc[e] = scale[cc * 8 + e]; sh[e] = shift[cc * 8 + e]; mu[e] = mean[cc * 8 + e]; rs[e] = rstd[cc * 8 + e];
    }
    const long stride = (long)gridDim.x * rpb;
    for (long m0 = (long)blockIdx.x * rpb + rl; m0 < M; m0 += stride * U) {
      bf16x8 g[U], v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long m = m0 + u * stride;
        if (m < M) {
          g[u] = *reinterpret_cast<const bf16x8*>(dA + m * lda + dacoff + cc * 8);
          v[u] = *reinterpret_cast<const bf16x8*>(y + m * ldy + cc * 8);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (m0 + u * stride >= M) break;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float yv = (float)v[u][e];
          const float z = __builtin_fmaf(yv, sc[e], sh[e]);
          const float sg = kod_sigmoid_l2(KOD_NEG_LOG2E * z);
          const float dz = kod_silu_bwd((float)g[u][e], z, sg);
          s0[e] += dz;
          s1[e] += dz * (yv - mu[e]) * rs[e];
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sm[(rl * CC + cc) * 16 + e] = s0[e];
      sm[(rl * CC + cc) * 16 + 8 + e] = s1[e];
    }
  }
  __syncthreads();
  // fixed-order reduction over the row lanes
  for (int i = threadIdx.x; i < CC * 16; i += blockDim.x) {
    int c8 = i >> 4, e = i & 15;
    float s = 0.f;
    for (int r = 0; r < rpb; ++r) s += sm[(r * CC + c8) * 16 + e];
    int st = e >> 3, ch = c8 * 8 + (e & 7);
    part[((size_t)st * C + ch) * gridDim.x + blockIdx.x] = s;
  }
}

// ---------------------------------------------------------------- activations other than SiLU
// The reference's layers take any activation callable (kod/nn/layers/csp.py:16-46, sppf.py:14-27, networks/yolov5.py:40-50);
// its configs only ever use SiLUInplace, which the tuned kernels above (and the fused epilogues in conv_igemm.hip /
// conv_wgrad.hip) implement.  These three plain passes carry the other elementwise activations torch offers for the slot -
// ReLU, LeakyReLU(slope), Hardswish, Identity (activation_layer=None) - with torch's conventions at the kinks
// (aten/native/cpu/Activation.cpp: relu' (0) = 0, leaky_relu' uses x > 0, hardswish' = 0 up to -3, x / 3 + 0.5 inside (-3, 3), 1 from 3 on);
// a network built with one of them runs its BatchNorm-backward reduction as its own pass (no fused epilogues).
enum { ACT_SILU = 0, ACT_RELU = 1, ACT_LEAKY = 2, ACT_HARDSWISH = 3, ACT_IDENTITY = 4 };
template <int ACT> __device__ __forceinline__ float kod_act(float z, float slope) {
  if (ACT == ACT_RELU) return z > 0.f ? z : 0.f;
  if (ACT == ACT_LEAKY) return z > 0.f ? z : z * slope;
  if (ACT == ACT_HARDSWISH) return z * fminf(fmaxf(z + 3.f, 0.f), 6.f) / 6.f;
  return z;
}
template <int ACT> __device__ __forceinline__ float kod_act_bwd(float g, float z, float slope) {
  if (ACT == ACT_RELU) return z > 0.f ? g : 0.f;
  if (ACT == ACT_LEAKY) return z > 0.f ? g : g * slope;
  if (ACT == ACT_HARDSWISH) return z <= -3.f ? 0.f : (z < 3.f ? g * ((z / 3.f) + 0.5f) : g);     // (torch 2.x: 0 at -3, g at 3)
  return g;
}

template <int ACT>
__global__ __launch_bounds__(256) void bn_act_apply_kernel(const bf16_t* y, int ldy, const float* scale, const float* shift,
                                                           const bf16_t* res, int ldr, int rcoff, bf16_t* out, int ldo, int ocoff,
                                                           long M, int C, int rpb, float slope) {
  const int CC = C >> 3, cc = threadIdx.x % CC, rl = threadIdx.x / CC;
  if (rl >= rpb) return;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sc[e] = scale[cc * 8 + e]; sh[e] = shift[cc * 8 + e]; }
  for (long m = (long)blockIdx.x * rpb + rl; m < M; m += (long)gridDim.x * rpb) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(y + m * ldy + cc * 8);
    bf16x8 r = {}, o;
    if (res) r = *reinterpret_cast<const bf16x8*>(res + m * ldr + rcoff + cc * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float a = kod_act<ACT>(__builtin_fmaf((float)v[e], sc[e], sh[e]), slope);
      o[e] = (bf16_t)(res ? a + (float)r[e] : a);
    }
    *reinterpret_cast<bf16x8*>(out + m * ldo + ocoff + cc * 8) = o;
  }
}

template <int ACT>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(const bf16_t* dA, int lda, int dacoff, bf16_t* y, int ldy,
                                                               const float* scale, const float* shift, const float* coef,
                                                               bf16_t* dI, int ldi, int dicoff, int di_accum, long M, int C, int rpb,
                                                               float slope) {
  const int CC = C >> 3, cc = threadIdx.x % CC, rl = threadIdx.x / CC;
  if (rl >= rpb) return;
  float sc[8], sh[8], k1[8], k2[8], k3[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = cc * 8 + e;
    sc[e] = scale[c]; sh[e] = shift[c]; k1[e] = coef[c]; k2[e] = coef[C + c]; k3[e] = coef[2 * C + c];
  }
  for (long m = (long)blockIdx.x * rpb + rl; m < M; m += (long)gridDim.x * rpb) {
    const bf16x8 g = *reinterpret_cast<const bf16x8*>(dA + m * lda + dacoff + cc * 8);
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(y + m * ldy + cc * 8);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float yv = (float)v[e];
      const float dz = kod_act_bwd<ACT>((float)g[e], __builtin_fmaf(yv, sc[e], sh[e]), slope);
      o[e] = (bf16_t)__builtin_fmaf(k1[e], dz, __builtin_fmaf(k2[e], yv, k3[e]));
    }
    *reinterpret_cast<bf16x8*>(y + m * ldy + cc * 8) = o;
    if (dI) {
      bf16x8 gi = g;
      if (di_accum) {
        const bf16x8 old = *reinterpret_cast<const bf16x8*>(dI + m * ldi + dicoff + cc * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) gi[e] = (bf16_t)((float)gi[e] + (float)old[e]);
      }
      *reinterpret_cast<bf16x8*>(dI + m * ldi + dicoff + cc * 8) = gi;
    }
  }
}

template <int ACT>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(const bf16_t* dA, int lda, int dacoff, const bf16_t* y, int ldy,
                                                                const float* scale, const float* shift, const float* mean,
                                                                const float* rstd, float* part, long M, int C, int rpb, float slope) {
  extern __shared__ float sm[];   // [rpb][CC][16]
  const int CC = C >> 3, cc = threadIdx.x % CC, rl = threadIdx.x / CC;
  float s0[8], s1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s0[e] = s1[e] = 0.f;
  if (rl < rpb) {
    float sc[8], sh[8], mu[8], rs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = scale[cc * 8 + e]; sh[e] = shift[cc * 8 + e]; mu[e] = mean[cc * 8 + e]; rs[e] = rstd[cc * 8 + e]; }
    for (long m = (long)blockIdx.x * rpb + rl; m < M; m += (long)gridDim.x * rpb) {
      const bf16x8 g = *reinterpret_cast<const bf16x8*>(dA + m * lda + dacoff + cc * 8);
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(y + m * ldy + cc * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float yv = (float)v[e];
        const float dz = kod_act_bwd<ACT>((float)g[e], __builtin_fmaf(yv, sc[e], sh[e]), slope);
        s0[e] += dz;
        s1[e] += dz * (yv - mu[e]) * rs[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { sm[(rl * CC + cc) * 16 + e] = s0[e]; sm[(rl * CC + cc) * 16 + 8 + e] = s1[e]; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < CC * 16; i += blockDim.x) {          // fixed-order reduction over the row lanes
    const int c8 = i >> 4, e = i & 15;
    float s = 0.f;
    for (int r = 0; r < rpb; ++r) s += sm[(r * CC + c8) * 16 + e];
    part[((size_t)(e >> 3) * C + c8 * 8 + (e & 7)) * gridDim.x + blockIdx.x] = s;
  }
}

// grads from the LOCAL sums; coefficients from the (all-reduced) sums:
//   dY = k1*dz + k2*y + k3,  k1 = g*rstd, k2 = -g*rstd^2*S1/n, k3 = -g*rstd*S0/n + g*rstd^2*mean*S1/n
__global__ void bn_bwd_coeffs_kernel(const double* sums_local, const double* sums_global, double count,
                                     const float* gamma, const float* mean, const float* rstd,
                                     float* dgamma, float* dbeta, float* coef, int C, int raw_moment) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double g = gamma[c], rs = rstd[c], mu = mean[c];
  double l1 = sums_local[C + c], g1 = sums_global[C + c];
  if (raw_moment) { l1 = rs * (l1 - mu * sums_local[c]); g1 = rs * (g1 - mu * sums_global[c]); }
  dbeta[c] = (float)sums_local[c];
  dgamma[c] = (float)l1;
  double S0 = sums_global[c] / count, S1 = g1 / count;
  coef[c] = (float)(g * rs);
  coef[C + c] = (float)(-g * rs * rs * S1);
  coef[2 * C + c] = (float)(-g * rs * S0 + g * rs * rs * mu * S1);
}

// dY (bf16, written in place over y) ; optional identity gradient: dI[m][c] (+)= dA[m][c]
template <int U, bool LDSK, int MAXT>
__global__ __launch_bounds__(MAXT) void bn_silu_bwd_apply_kernel(const bf16_t* dA, int lda, int dacoff, bf16_t* y, int ldy,
                                         const float* scale, const float* shift, const float* coef,
                                         bf16_t* dI, int ldi, int dicoff, int di_accum,
                                         long M, int C, int rpb) {
  extern __shared__ float kconst[];           // LDSK: [scale C | shift C | k1 C | k2 C | k3 C]
  const int CC = C >> 3;
  const int cc = threadIdx.x % CC;
  const int rl = threadIdx.x / CC;
  float sc[8], sh[8], k1[8], k2[8], k3[8];
  if (LDSK) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      kconst[c] = scale[c]; kconst[C + c] = shift[c];
      kconst[2 * C + c] = coef[c]; kconst[3 * C + c] = coef[C + c]; kconst[4 * C + c] = coef[2 * C + c];
    }
    __syncthreads();
  }
  if (rl >= rpb) return;
  if (!LDSK) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int c = cc * 8 + e;
      sc[e] = scale[c]; sh[e] = shift[c]; k1[e] = coef[c]; k2[e] = coef[C + c]; k3[e] = coef[2 * C + c];
    }
  }
  int ko = cc * 8;
  const long step = (long)gridDim.x * rpb * U;
  const bool acc = dI && di_accum;
  for (long m0 = (long)blockIdx.x * rpb * U + rl; m0 < M; m0 += step) {
    bf16x8 g[U], v[U], old[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long m = m0 + (long)u * rpb;
      if (m < M) {
        g[u] = kod_load_once<bf16x8>(dA + m * lda + dacoff + cc * 8);
        v[u] = kod_load_once<bf16x8>(y + m * ldy + cc * 8);
        if (acc) old[u] = *reinterpret_cast<const bf16x8*>(dI + m * ldi + dicoff + cc * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long m = m0 + (long)u * rpb;
      if (m >= M) break;
      bf16x8 o;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (LDSK) {
          // the forty per-thread constants are read from LDS where they are used, four channels at a time, instead of being
          // held in registers across the loop: 8 waves per SIMD instead of 6 keep enough bytes in flight for the HBM latency
          // (the opaque offset keeps the compiler from hoisting the reads back out of the loop)
          asm volatile("" : "+v"(ko));
          const f32x4 a0 = *reinterpret_cast<const f32x4*>(&kconst[ko + 4 * h]), a1 = *reinterpret_cast<const f32x4*>(&kconst[C + ko + 4 * h]);
          const f32x4 a2 = *reinterpret_cast<const f32x4*>(&kconst[2 * C + ko + 4 * h]), a3 = *reinterpret_cast<const f32x4*>(&kconst[3 * C + ko + 4 * h]);
          const f32x4 a4 = *reinterpret_cast<const f32x4*>(&kconst[4 * C + ko + 4 * h]);
#pragma unroll
          for (int e = 0; e < 4; ++e) { sc[4 * h + e] = a0[e]; sh[4 * h + e] = a1[e]; k1[4 * h + e] = a2[e]; k2[4 * h + e] = a3[e]; k3[4 * h + e] = a4[e]; }
        }
#pragma unroll
        for (int e = 4 * h; e < 4 * h + 4; ++e) {
          const float yv = (float)v[u][e];
          const float z = __builtin_fmaf(yv, sc[e], sh[e]);
          const float sg = kod_sigmoid_l2(KOD_NEG_LOG2E * z);
          const float dz = kod_silu_bwd((float)g[u][e], z, sg);
          o[e] = (bf16_t)__builtin_fmaf(k1[e], dz, __builtin_fmaf(k2[e], yv, k3[e]));
        }
      }
      *reinterpret_cast<bf16x8*>(y + m * ldy + cc * 8) = o;
      if (dI) {
        bf16x8 gi = g[u];
        if (acc) {
#pragma unroll
          for (int e = 0; e < 8; ++e) gi[e] = (bf16_t)((float)gi[e] + (float)old[u][e]);
        }
        *reinterpret_cast<bf16x8*>(dI + m * ldi + dicoff + cc * 8) = gi;
      }
    }
  }
}

// Launch shape of the two apply passes.  Round 5 (profiles/r05_bn_apply_sweep.txt): one row per thread and many blocks beat four
// grid-strided rows per thread.  Round 6, on tensors as cold as they are inside a step (tools/micro/stream_apply.hip,
// profiles/r06_stream_apply.txt; tools/bench_bn.py, profiles/r06_bn_apply_sweep.txt): the passes were bounded by the bytes in
// flight, and what held those down was the register cost of the per-thread constants (74 registers = 6 waves per SIMD in the
// backward pass) and, with short blocks, the constant loads outnumbering the payload's.  Backward, <= 64 channels: constants in
// LDS, read where they are used (50 registers), one row per thread, one chunk per block - 419 MB 235 -> 188 us (5.4 -> 6.7 TB/s),
// 210 MB 114 -> 96 us; wider tensors have too few rows per block to pay for staging the constants and take two
// block-adjacent rows per thread with the constants in registers (59 -> 53 us at 409 600 x 128).  Forward (16 constants, 62
// registers either way): two block-adjacent rows per thread, at most 32 768 blocks: - 5 % over the nine sizes.
struct ApplyShape { int threads, u, lds, cap; };
static int bn_knob(const char* name) { const char* e = getenv(name); return e ? atoi(e) : 0; }
static ApplyShape apply_shape(long M, int C, bool backward) {
  // A/B knobs, read once (thread-safe: function-local static initialisers); 0 / unset = the measured default
  static const int ku = bn_knob("KODHIP_BN_U"), kg = bn_knob("KODHIP_BN_GRID"), kb = bn_knob("KODHIP_BN_BLOCK"), kl = bn_knob("KODHIP_BN_LDS");
  (void)M;
  ApplyShape a;
  if (backward && C <= 64) { a.lds = 1; a.u = 1; a.threads = 256; a.cap = 1 << 30; }
  else { a.lds = 0; a.u = 2; a.threads = 256; a.cap = 32768; }
  if (ku == 1 || ku == 2 || ku == 4) a.u = ku;
  if (kb == 256 || kb == 512 || kb == 1024) a.threads = kb;
  if (kl) a.lds = kl > 0 ? 1 : 0;          // KODHIP_BN_LDS = 1 | -1
  if (kg) a.cap = kg;
  if (a.threads > 256 && a.u > 2) a.u = 2;  // four rows in flight need more than the 128 registers a 1024-thread block may use
  return a;
}

struct Geo { int threads, rpb, grid; };
Geo geo(long M, int C, int max_blocks, int max_threads = 256, int U = 1) {
  int CC = C / 8;
  int rpb = max_threads / CC;
  // whole 128-byte lines per block chunk where the width allows it (48 channels: 40 rows = 3 840 B instead of 42 = 4 032 B)
  int q = 1;
  while ((q * C * 2) % 128) q *= 2;
  if (rpb >= q) rpb -= rpb % q;
  if (rpb < 1) rpb = 1;
  Geo g;
  g.rpb = rpb;
  g.threads = ((rpb * CC + 63) / 64) * 64;
  long blocks = (M + (long)rpb * U - 1) / ((long)rpb * U);
  g.grid = (int)(blocks < max_blocks ? blocks : max_blocks);
  if (g.grid < 1) g.grid = 1;
  return g;
}

// (rows in flight, constants through LDS, block size class) -> template instance; `a` and `g` in scope
#define KOD_APPLY_DISPATCH(L)                                                                              \
  do {                                                                                                     \
    const bool big = g.threads > 256;                                                                      \
    if (a.u == 4) { if (a.lds) L(4, true, 256); else L(4, false, 256); }                                   \
    else if (a.u == 2) { if (big) { if (a.lds) L(2, true, 1024); else L(2, false, 1024); }                 \
                         else { if (a.lds) L(2, true, 256); else L(2, false, 256); } }                     \
    else { if (big) { if (a.lds) L(1, true, 1024); else L(1, false, 1024); }                               \
           else { if (a.lds) L(1, true, 256); else L(1, false, 256); } }                                   \
  } while (0)

}  // namespace

extern "C" {

int kodhip_bn_reduce_partials(const float* partials, double* sums, int C, int T, hipStream_t stream) {
  KOD_CHECK_ARG(partials && sums && C > 0 && T > 0, "bn_reduce_partials: bad args");
  hipLaunchKernelGGL(bn_reduce_partials_kernel, dim3(cdiv(2 * C, 4)), dim3(256), 0, stream, partials, sums, C, T);
  KOD_LAUNCH_CHECK("bn_reduce_partials");
  return KOD_OK;
}

int kodhip_bn_finalize(const double* sums, double count, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, float momentum, float eps,
                       float* scale, float* shift, float* mean, float* rstd, int C, int update_running,
                       hipStream_t stream) {
  KOD_CHECK_ARG(sums && gamma && beta && scale && shift && mean && rstd && C > 0 && count > 0, "bn_finalize: bad args");
  KOD_CHECK_ARG(!update_running || (running_mean && running_var), "bn_finalize: running stats missing");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 64)), dim3(64), 0, stream, sums, count, gamma, beta,
                     running_mean, running_var, momentum, eps, scale, shift, mean, rstd, C, update_running);
  KOD_LAUNCH_CHECK("bn_finalize");
  return KOD_OK;
}

int kodhip_bn_finalize_partials(const float* partials, int T, double count, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, float momentum, float eps,
                                float* scale, float* shift, float* mean, float* rstd, int C, int update_running,
                                hipStream_t stream) {
  KOD_CHECK_ARG(partials && gamma && beta && scale && shift && mean && rstd && C > 0 && T > 0 && count > 0,
                "bn_finalize_partials: bad args");
  KOD_CHECK_ARG(!update_running || (running_mean && running_var), "bn_finalize_partials: running stats missing");
  hipLaunchKernelGGL(bn_finalize_fused_kernel<false>, dim3(cdiv(C, 4)), dim3(256), 0, stream, partials, T, count, gamma, beta,
                     running_mean, running_var, momentum, eps, scale, shift, mean, rstd, C, update_running, KodPeerView{}, 0u);
  KOD_LAUNCH_CHECK("bn_finalize_partials");
  return KOD_OK;
}

// Batch statistics -> BatchNorm constants of TWO units whose convolution was one launch with N = 2 * Ch columns
// (partials [2][2 * Ch][T], unit h owns channels [h * Ch, (h + 1) * Ch)): one launch for both.  aff0 / aff1 =
// scale | shift | mean | rstd (4 * Ch floats each).  view != NULL: SyncBN over the peer buffers (slot0 / slot1 = first
// granule of each unit's exchange, count = pixels of ALL ranks).
int kodhip_bn_finalize_partials_pair(const float* partials, int T, double count, int Ch, float momentum, float eps,
                                     int update_running,
                                     const float* gamma0, const float* beta0, float* running_mean0, float* running_var0, float* aff0,
                                     const float* gamma1, const float* beta1, float* running_mean1, float* running_var1, float* aff1,
                                     const void* view, unsigned int slot0, unsigned int slot1, hipStream_t stream) {
  KOD_CHECK_ARG(partials && gamma0 && beta0 && aff0 && gamma1 && beta1 && aff1 && Ch > 0 && T > 0 && count > 0,
                "bn_finalize_partials_pair: bad args");
  KOD_CHECK_ARG(!update_running || (running_mean0 && running_var0 && running_mean1 && running_var1),
                "bn_finalize_partials_pair: running stats missing");
  FinJob j0 = {gamma0, beta0, running_mean0, running_var0, aff0, aff0 + Ch, aff0 + 2 * Ch, aff0 + 3 * Ch, slot0};
  FinJob j1 = {gamma1, beta1, running_mean1, running_var1, aff1, aff1 + Ch, aff1 + 2 * Ch, aff1 + 3 * Ch, slot1};
  if (view) {
    const KodPeerView pv = *(const KodPeerView*)view;
    KOD_CHECK_ARG(pv.world >= 1 && pv.world <= KOD_PEER_MAX && pv.world * 4 <= 64, "bn_finalize_partials_pair: bad peer view");
    hipLaunchKernelGGL(bn_finalize_pair_kernel<true>, dim3(cdiv(Ch, 4), 2), dim3(256), 0, stream, partials, T, count, momentum,
                       eps, Ch, update_running, j0, j1, pv);
  } else {
    hipLaunchKernelGGL(bn_finalize_pair_kernel<false>, dim3(cdiv(Ch, 4), 2), dim3(256), 0, stream, partials, T, count, momentum,
                       eps, Ch, update_running, j0, j1, KodPeerView{});
  }
  KOD_LAUNCH_CHECK("bn_finalize_partials_pair");
  return KOD_OK;
}

// SyncBN forms (kod/configs/trainer/ddp.yaml:9 sync_batchnorm): the same single launches, with the rank's two sums
// per channel exchanged through the peer buffers of kodhip_peer_* inside the kernel.  count = pixels of ALL ranks;
// slot = first granule of this exchange in the buffers (4 * C granules); view = kodhip_peer_view's struct.
int kodhip_bn_finalize_partials_peer(const float* partials, int T, double count, const float* gamma, const float* beta,
                                     float* running_mean, float* running_var, float momentum, float eps,
                                     float* scale, float* shift, float* mean, float* rstd, int C, int update_running,
                                     const void* view, unsigned int slot, hipStream_t stream) {
  KOD_CHECK_ARG(partials && gamma && beta && scale && shift && mean && rstd && C > 0 && T > 0 && count > 0 && view,
                "bn_finalize_partials_peer: bad args");
  KOD_CHECK_ARG(!update_running || (running_mean && running_var), "bn_finalize_partials_peer: running stats missing");
  const KodPeerView pv = *(const KodPeerView*)view;
  KOD_CHECK_ARG(pv.world >= 1 && pv.world <= KOD_PEER_MAX && pv.world * 4 <= 64, "bn_finalize_partials_peer: bad peer view");
  hipLaunchKernelGGL(bn_finalize_fused_kernel<true>, dim3(cdiv(C, 4)), dim3(256), 0, stream, partials, T, count, gamma, beta,
                     running_mean, running_var, momentum, eps, scale, shift, mean, rstd, C, update_running, pv, slot);
  KOD_LAUNCH_CHECK("bn_finalize_partials_peer");
  return KOD_OK;
}

int kodhip_bn_bwd_coeffs_partials_peer(const float* partials, int T, double count, const float* gamma, const float* mean,
                                       const float* rstd, float* dgamma, float* dbeta, float* coef, int C,
                                       int raw_moment, const void* view, unsigned int slot, hipStream_t stream) {
  KOD_CHECK_ARG(partials && gamma && mean && rstd && dgamma && dbeta && coef && C > 0 && T > 0 && count > 0 && view,
                "bn_bwd_coeffs_partials_peer: bad args");
  const KodPeerView pv = *(const KodPeerView*)view;
  KOD_CHECK_ARG(pv.world >= 1 && pv.world <= KOD_PEER_MAX, "bn_bwd_coeffs_partials_peer: bad peer view");
  hipLaunchKernelGGL(bn_bwd_coeffs_peer_kernel, dim3(cdiv(C, 4)), dim3(256), 0, stream, partials, T, count, gamma,
                     mean, rstd, dgamma, dbeta, coef, C, raw_moment, pv, slot);
  KOD_LAUNCH_CHECK("bn_bwd_coeffs_partials_peer");
  return KOD_OK;
}

int kodhip_bn_bwd_coeffs_partials(const float* partials, int T, double count, const float* gamma, const float* mean,
                                  const float* rstd, float* dgamma, float* dbeta, float* coef, int C,
                                  int raw_moment, hipStream_t stream) {
  KOD_CHECK_ARG(partials && gamma && mean && rstd && dgamma && dbeta && coef && C > 0 && T > 0 && count > 0,
                "bn_bwd_coeffs_partials: bad args");
  hipLaunchKernelGGL(bn_bwd_coeffs_fused_kernel, dim3(cdiv(C, 4)), dim3(256), 0, stream, partials, T, count, gamma,
                     mean, rstd, dgamma, dbeta, coef, C, raw_moment);
  KOD_LAUNCH_CHECK("bn_bwd_coeffs_partials");
  return KOD_OK;
}

int kodhip_bn_bwd_coeffs_partials2(const float* partials0, int T0, double count0, const float* gamma0, const float* mean0,
                                   const float* rstd0, float* dgamma0, float* dbeta0, float* coef0, int C0, int raw_moment0,
                                   const float* partials1, int T1, double count1, const float* gamma1, const float* mean1,
                                   const float* rstd1, float* dgamma1, float* dbeta1, float* coef1, int C1, int raw_moment1,
                                   hipStream_t stream) {
  KOD_CHECK_ARG(partials0 && gamma0 && mean0 && rstd0 && dgamma0 && dbeta0 && coef0 && C0 > 0 && T0 > 0 && count0 > 0 &&
                partials1 && gamma1 && mean1 && rstd1 && dgamma1 && dbeta1 && coef1 && C1 > 0 && T1 > 0 && count1 > 0,
                "bn_bwd_coeffs_partials2: bad args");
  CoefJob j0 = {partials0, T0, count0, gamma0, mean0, rstd0, dgamma0, dbeta0, coef0, C0, raw_moment0};
  CoefJob j1 = {partials1, T1, count1, gamma1, mean1, rstd1, dgamma1, dbeta1, coef1, C1, raw_moment1};
  hipLaunchKernelGGL(bn_bwd_coeffs_fused2_kernel, dim3(cdiv(C0 > C1 ? C0 : C1, 4), 2), dim3(256), 0, stream, j0, j1);
  KOD_LAUNCH_CHECK("bn_bwd_coeffs_partials2");
  return KOD_OK;
}

int kodhip_bn_silu_apply(const void* y, int ldy, const float* scale, const float* shift,
                         const void* residual, int ldr, int rcoff,
                         void* out, int ldo, int ocoff, long M, int C, hipStream_t stream) {
  KOD_CHECK_ARG(y && scale && shift && out && M > 0, "bn_silu_apply: bad args");
  KOD_CHECK_ARG(C % 8 == 0 && C <= 2048 && ldo % 8 == 0 && ocoff % 8 == 0 && ocoff + C <= ldo, "bn_silu_apply: bad channel geometry");
  KOD_CHECK_ARG(!residual || (ldr % 8 == 0 && rcoff % 8 == 0 && rcoff + C <= ldr), "bn_silu_apply: bad residual slice");
  const ApplyShape a = apply_shape(M, C, false);
  Geo g = geo(M, C, a.cap, a.threads, a.u);
  KOD_CHECK_ARG(ldy % 8 == 0 && ldy >= C, "bn_silu_apply: bad row stride of y");
#define KOD_APPLY(UU, LL, TT) hipLaunchKernelGGL((bn_silu_apply_kernel<UU, LL, TT>), dim3(g.grid), dim3(g.threads), LL ? 2 * C * sizeof(float) : 0, stream, \
                     (const bf16_t*)y, ldy, scale, shift, (const bf16_t*)residual, ldr, rcoff, (bf16_t*)out, ldo, ocoff, M, C, g.rpb)
  KOD_APPLY_DISPATCH(KOD_APPLY);
#undef KOD_APPLY
  KOD_LAUNCH_CHECK("bn_silu_apply");
  return KOD_OK;
}

// y[m][2 * Ch] (row stride ldy) -> silu(bn(.)) of channels [0, Ch) into out0's slice and of [Ch, 2 * Ch) into out1's:
// the two apply passes of a CSP layer's main_conv / short_conv pair as one launch (same arithmetic as kodhip_bn_silu_apply)
int kodhip_bn_silu_apply_pair(const void* y, int ldy, int Ch,
                              const float* scale0, const float* shift0, void* out0, int ldo0, int ocoff0,
                              const float* scale1, const float* shift1, void* out1, int ldo1, int ocoff1,
                              long M, hipStream_t stream) {
  KOD_CHECK_ARG(y && scale0 && shift0 && out0 && scale1 && shift1 && out1 && M > 0, "bn_silu_apply_pair: bad args");
  KOD_CHECK_ARG(Ch % 8 == 0 && Ch > 0 && 2 * Ch <= 2048 && ldy % 8 == 0 && ldy >= 2 * Ch, "bn_silu_apply_pair: bad channel geometry");
  KOD_CHECK_ARG(ldo0 % 8 == 0 && ocoff0 % 8 == 0 && ocoff0 + Ch <= ldo0 && ldo1 % 8 == 0 && ocoff1 % 8 == 0 && ocoff1 + Ch <= ldo1,
                "bn_silu_apply_pair: bad destination slice");
  Geo g = geo(M, 2 * Ch, 4096);
  hipLaunchKernelGGL(bn_silu_apply_pair_kernel, dim3(g.grid), dim3(g.threads), 0, stream, (const bf16_t*)y, ldy,
                     scale0, shift0, (bf16_t*)out0, ldo0, ocoff0, scale1, shift1, (bf16_t*)out1, ldo1, ocoff1, M, Ch, g.rpb);
  KOD_LAUNCH_CHECK("bn_silu_apply_pair");
  return KOD_OK;
}

// Reduction grid of the backward pass = number of partial slots: at most 1024 blocks, and no more than one block per
// 64 KB of input, so that small layers do not pay for a partial slab (2*C*grid floats) as large as their data.
static int bwd_reduce_blocks(long M, int C) {
  long by_bytes = (M * C * 4 + 65535) / 65536;
  if (by_bytes < 64) by_bytes = 64;
  return (int)(by_bytes < 1024 ? by_bytes : 1024);
}

int kodhip_bn_bwd_slots(long M, int C) { return geo(M, C, bwd_reduce_blocks(M, C)).grid; }

int kodhip_bn_silu_bwd_reduce(const void* dA, int lda, int dacoff, const void* y, int ldy, const float* scale,
                              const float* shift, const float* mean, const float* rstd, float* partials,
                              long M, int C, hipStream_t stream) {
  KOD_CHECK_ARG(dA && y && scale && shift && mean && rstd && partials && M > 0, "bn_silu_bwd_reduce: bad args");
  KOD_CHECK_ARG(C % 8 == 0 && C <= 2048 && lda % 8 == 0 && dacoff % 8 == 0 && dacoff + C <= lda, "bn_silu_bwd_reduce: bad geometry");
  Geo g = geo(M, C, bwd_reduce_blocks(M, C));
  size_t shm = (size_t)g.rpb * (C / 8) * 16 * sizeof(float);
  KOD_CHECK_ARG(ldy % 8 == 0 && ldy >= C, "bn_silu_bwd_reduce: bad row stride of y");
  hipLaunchKernelGGL(bn_silu_bwd_reduce_kernel, dim3(g.grid), dim3(g.threads), shm, stream, (const bf16_t*)dA, lda,
                     dacoff, (const bf16_t*)y, ldy, scale, shift, mean, rstd, partials, M, C, g.rpb);
  KOD_LAUNCH_CHECK("bn_silu_bwd_reduce");
  return KOD_OK;
}

int kodhip_bn_bwd_coeffs(const double* sums_local, const double* sums_global, double count, const float* gamma,
                         const float* mean, const float* rstd, float* dgamma, float* dbeta, float* coef, int C,
                         int raw_moment, hipStream_t stream) {
  KOD_CHECK_ARG(sums_local && sums_global && gamma && mean && rstd && dgamma && dbeta && coef && C > 0 && count > 0,
                "bn_bwd_coeffs: bad args");
  hipLaunchKernelGGL(bn_bwd_coeffs_kernel, dim3(cdiv(C, 64)), dim3(64), 0, stream, sums_local, sums_global, count,
                     gamma, mean, rstd, dgamma, dbeta, coef, C, raw_moment);
  KOD_LAUNCH_CHECK("bn_bwd_coeffs");
  return KOD_OK;
}

int kodhip_bn_silu_bwd_apply(const void* dA, int lda, int dacoff, void* y_inout, int ldy, const float* scale,
                             const float* shift, const float* coef, void* dI, int ldi, int dicoff, int di_accum,
                             long M, int C, hipStream_t stream) {
  KOD_CHECK_ARG(dA && y_inout && scale && shift && coef && M > 0, "bn_silu_bwd_apply: bad args");
  KOD_CHECK_ARG(C % 8 == 0 && C <= 2048 && lda % 8 == 0 && dacoff % 8 == 0 && dacoff + C <= lda, "bn_silu_bwd_apply: bad geometry");
  KOD_CHECK_ARG(!dI || (ldi % 8 == 0 && dicoff % 8 == 0 && dicoff + C <= ldi), "bn_silu_bwd_apply: bad identity slice");
  const ApplyShape a = apply_shape(M, C, true);
  Geo g = geo(M, C, a.cap, a.threads, a.u);
  KOD_CHECK_ARG(ldy % 8 == 0 && ldy >= C, "bn_silu_bwd_apply: bad row stride of y");
#define KOD_BAPPLY(UU, LL, TT) hipLaunchKernelGGL((bn_silu_bwd_apply_kernel<UU, LL, TT>), dim3(g.grid), dim3(g.threads), LL ? 5 * C * sizeof(float) : 0, stream, \
                     (const bf16_t*)dA, lda, dacoff, (bf16_t*)y_inout, ldy, scale, shift, coef, (bf16_t*)dI, ldi, dicoff, di_accum, M, C, g.rpb)
  KOD_APPLY_DISPATCH(KOD_BAPPLY);
#undef KOD_BAPPLY
  KOD_LAUNCH_CHECK("bn_silu_bwd_apply");
  return KOD_OK;
}

// ---- the same three passes for an activation other than SiLU (see bn_act_apply_kernel): act = 0 SiLU (the tuned kernels
// above), 1 ReLU, 2 LeakyReLU(slope), 3 Hardswish, 4 identity (activation_layer=None).  Arguments as in the SiLU entries.
#define KOD_ACT_DISPATCH(KERNEL, ...)                                                             \
  switch (act) {                                                                                  \
    case ACT_RELU: hipLaunchKernelGGL(KERNEL<ACT_RELU>, __VA_ARGS__); break;                      \
    case ACT_LEAKY: hipLaunchKernelGGL(KERNEL<ACT_LEAKY>, __VA_ARGS__); break;                    \
    case ACT_HARDSWISH: hipLaunchKernelGGL(KERNEL<ACT_HARDSWISH>, __VA_ARGS__); break;            \
    default: hipLaunchKernelGGL(KERNEL<ACT_IDENTITY>, __VA_ARGS__); break;                        \
  }

int kodhip_bn_act_apply(const void* y, int ldy, const float* scale, const float* shift, const void* residual, int ldr, int rcoff,
                        void* out, int ldo, int ocoff, long M, int C, int act, float slope, hipStream_t stream) {
  if (act == ACT_SILU) return kodhip_bn_silu_apply(y, ldy, scale, shift, residual, ldr, rcoff, out, ldo, ocoff, M, C, stream);
  KOD_CHECK_ARG(act >= 1 && act <= 4, "bn_act_apply: activation code %d", act);
  KOD_CHECK_ARG(y && scale && shift && out && M > 0, "bn_act_apply: bad args");
  KOD_CHECK_ARG(C % 8 == 0 && C <= 2048 && ldo % 8 == 0 && ocoff % 8 == 0 && ocoff + C <= ldo && ldy % 8 == 0 && ldy >= C, "bn_act_apply: bad channel geometry");
  KOD_CHECK_ARG(!residual || (ldr % 8 == 0 && rcoff % 8 == 0 && rcoff + C <= ldr), "bn_act_apply: bad residual slice");
  Geo g = geo(M, C, (M * C * 2 >= (128l << 20)) ? 16384 : 4096);
  KOD_ACT_DISPATCH(bn_act_apply_kernel, dim3(g.grid), dim3(g.threads), 0, stream, (const bf16_t*)y, ldy, scale, shift, (const bf16_t*)residual,
                   ldr, rcoff, (bf16_t*)out, ldo, ocoff, M, C, g.rpb, slope)
  KOD_LAUNCH_CHECK("bn_act_apply");
  return KOD_OK;
}

int kodhip_bn_act_bwd_apply(const void* dA, int lda, int dacoff, void* y_inout, int ldy, const float* scale, const float* shift,
                            const float* coef, void* dI, int ldi, int dicoff, int di_accum, long M, int C, int act, float slope,
                            hipStream_t stream) {
  if (act == ACT_SILU) return kodhip_bn_silu_bwd_apply(dA, lda, dacoff, y_inout, ldy, scale, shift, coef, dI, ldi, dicoff, di_accum, M, C, stream);
  KOD_CHECK_ARG(act >= 1 && act <= 4, "bn_act_bwd_apply: activation code %d", act);
  KOD_CHECK_ARG(dA && y_inout && scale && shift && coef && M > 0, "bn_act_bwd_apply: bad args");
  KOD_CHECK_ARG(C % 8 == 0 && C <= 2048 && lda % 8 == 0 && dacoff % 8 == 0 && dacoff + C <= lda && ldy % 8 == 0 && ldy >= C, "bn_act_bwd_apply: bad geometry");
  KOD_CHECK_ARG(!dI || (ldi % 8 == 0 && dicoff % 8 == 0 && dicoff + C <= ldi), "bn_act_bwd_apply: bad identity slice");
  Geo g = geo(M, C, (M * C * 2 >= (128l << 20)) ? 16384 : 4096);
  KOD_ACT_DISPATCH(bn_act_bwd_apply_kernel, dim3(g.grid), dim3(g.threads), 0, stream, (const bf16_t*)dA, lda, dacoff, (bf16_t*)y_inout, ldy,
                   scale, shift, coef, (bf16_t*)dI, ldi, dicoff, di_accum, M, C, g.rpb, slope)
  KOD_LAUNCH_CHECK("bn_act_bwd_apply");
  return KOD_OK;
}

int kodhip_bn_act_bwd_reduce(const void* dA, int lda, int dacoff, const void* y, int ldy, const float* scale, const float* shift,
                             const float* mean, const float* rstd, float* partials, long M, int C, int act, float slope,
                             hipStream_t stream) {
  if (act == ACT_SILU) return kodhip_bn_silu_bwd_reduce(dA, lda, dacoff, y, ldy, scale, shift, mean, rstd, partials, M, C, stream);
  KOD_CHECK_ARG(act >= 1 && act <= 4, "bn_act_bwd_reduce: activation code %d", act);
  KOD_CHECK_ARG(dA && y && scale && shift && mean && rstd && partials && M > 0, "bn_act_bwd_reduce: bad args");
  KOD_CHECK_ARG(C % 8 == 0 && C <= 2048 && lda % 8 == 0 && dacoff % 8 == 0 && dacoff + C <= lda && ldy % 8 == 0 && ldy >= C, "bn_act_bwd_reduce: bad geometry");
  Geo g = geo(M, C, bwd_reduce_blocks(M, C));
  const size_t shm = (size_t)g.rpb * (C / 8) * 16 * sizeof(float);
  KOD_ACT_DISPATCH(bn_act_bwd_reduce_kernel, dim3(g.grid), dim3(g.threads), shm, stream, (const bf16_t*)dA, lda, dacoff, (const bf16_t*)y, ldy,
                   scale, shift, mean, rstd, partials, M, C, g.rpb, slope)
  KOD_LAUNCH_CHECK("bn_act_bwd_reduce");
  return KOD_OK;
}

}  // extern "C"
