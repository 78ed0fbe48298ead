// What a conv launch's tail costs when every block ends by adding its per-channel partial sums into ONE set of accumulators
// with agent-scope 64-bit integer atomics (order-independent => deterministic) instead of storing them into its own slot:
// `blocks` workgroups x 256 threads, thread t adds to accumulator t (two moments: t and 256 + t), accumulators padded to
// `stride` bytes.  A busy-wait of ~20 us precedes the tail so that the blocks arrive together, as in a real launch.
// Diagnostic only: hipcc --offload-arch=gfx950 -O3 -o tools/micro/atomic_tail tools/micro/atomic_tail.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>   // 0: nothing, 1: plain stores into the block's own slots, 2: u64 atomics, 3: two u64 atomics + the slot stores
__global__ __launch_bounds__(256) void tail(unsigned long long* acc, float* slots, int stride_q, int spin, int nblocks) {
  const int t = threadIdx.x;
  long long t0 = __builtin_amdgcn_s_memtime();
  float v = (float)t;
  while (__builtin_amdgcn_s_memtime() - t0 < spin) v = v * 1.0001f + 0.5f;
  const long long q0 = (long long)(v * 1048576.0f), q1 = (long long)(v * v * 4096.0f);
  if (MODE == 1 || MODE == 3) {
    slots[(size_t)t * nblocks + blockIdx.x] = v;
    slots[(size_t)(256 + t) * nblocks + blockIdx.x] = v * v;
  }
  if (MODE >= 2) {
    __hip_atomic_fetch_add(acc + (size_t)t * stride_q, (unsigned long long)q0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(acc + (size_t)(256 + t) * stride_q, (unsigned long long)q1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int MODE>
static float run(int blocks, unsigned long long* acc, float* slots, int stride_q) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int spin = 2000;      // s_memtime ticks at 100 MHz: 20 us
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(tail<MODE>, dim3(blocks), dim3(256), 0, 0, acc, slots, stride_q, spin, blocks);
  (void)hipEventRecord(e0);
  for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(tail<MODE>, dim3(blocks), dim3(256), 0, 0, acc, slots, stride_q, spin, blocks);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / 20;
}

int main() {
  unsigned long long* acc; float* slots;
  (void)hipMalloc(&acc, 512 * 256); (void)hipMalloc(&slots, 512 * 2048 * 4);
  (void)hipMemset(acc, 0, 512 * 256);
  for (int blocks : {256, 512, 1024, 2048}) {
    const float base = run<0>(blocks, acc, slots, 1);
    printf("blocks %4d: no tail %6.2f us | slot stores %+6.2f | atomics, packed (8 B apart) %+6.2f | 64 B apart %+6.2f | 128 B apart %+6.2f | 256 B apart %+6.2f\n",
           blocks, base, run<1>(blocks, acc, slots, 1) - base, run<2>(blocks, acc, slots, 1) - base, run<2>(blocks, acc, slots, 8) - base,
           run<2>(blocks, acc, slots, 16) - base, run<2>(blocks, acc, slots, 32) - base);
  }
  return 0;
}
