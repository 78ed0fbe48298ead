// Design space of a 2-read + 1-write streaming pass (the BatchNorm / SiLU backward apply) on gfx950, outside the library:
// rows in flight per thread (adjacent or grid-strided), non-temporal or plain loads / stores, with and without the
// sigmoid arithmetic, in place or to a third buffer, block size, grid cap.  Prints us and GB/s (6 bytes per element).
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/stream_apply tools/micro/stream_apply.hip && tools/micro/stream_apply
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16_t;
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NT> __device__ __forceinline__ bf16x8 ld(const bf16_t* p) {
  if (NT & 1) return __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p));
  return *reinterpret_cast<const bf16x8*>(p);
}
template <int NT> __device__ __forceinline__ void st(bf16_t* p, bf16x8 v) {
  if (NT & 2) __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(p));
  else *reinterpret_cast<bf16x8*>(p) = v;
}

// n16 = number of 16-byte vectors; C/8 vectors per row; one vector per thread per row in flight.
// MATH: 0 = add, 1 = sigmoid arithmetic with 24 per-thread constants, 2 = the library's 40 constants loaded per thread from
// global memory, 3 = the same 40 staged through LDS once per block
template <int U, int ADJ, int NT, int MATH>
__global__ __launch_bounds__(1024) void apply2r1w(const bf16_t* __restrict__ a, const bf16_t* b, bf16_t* o, const float* k, long n16, int CC) {
  const int cc = threadIdx.x % CC;
  float sc[8], sh[8], k1[8], k2[8], k3[8];
  if (MATH == 3) {
    __shared__ float ks[5 * 256];
    for (int i = threadIdx.x; i < 5 * CC * 8; i += blockDim.x) ks[i] = k[(i / (CC * 8)) * 256 + i % (CC * 8)];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sc[e] = ks[cc * 8 + e]; sh[e] = ks[CC * 8 + cc * 8 + e]; k1[e] = ks[2 * CC * 8 + cc * 8 + e];
      k2[e] = ks[3 * CC * 8 + cc * 8 + e]; k3[e] = ks[4 * CC * 8 + cc * 8 + e];
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sc[e] = k[cc * 8 + e]; sh[e] = k[256 + cc * 8 + e]; k1[e] = k[512 + cc * 8 + e];
      if (MATH == 2) { k2[e] = k[768 + cc * 8 + e]; k3[e] = k[1024 + cc * 8 + e]; }
    }
  }
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i0 = ADJ ? (long)blockIdx.x * blockDim.x * U + threadIdx.x : (long)blockIdx.x * blockDim.x + threadIdx.x; i0 < n16; i0 += stride * U) {
    bf16x8 g[U], v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = ADJ ? i0 + (long)u * blockDim.x : i0 + u * stride;
      if (i < n16) { if (MATH != 4) g[u] = ld<NT>(a + i * 8); v[u] = ld<NT>(b + i * 8); }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = ADJ ? i0 + (long)u * blockDim.x : i0 + u * stride;
      if (i >= n16) break;
      bf16x8 r;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float yv = (float)v[u][e];
        if (MATH == 4) {               // the forward pass: 1 read + 1 write, 16 constants
          const float z = __builtin_fmaf(yv, sc[e], sh[e]);
          r[e] = (bf16_t)(z * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z)));
        } else if (MATH) {
          const float z = __builtin_fmaf(yv, sc[e], sh[e]);
          const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z));
          const float t = __builtin_fmaf(-z, sg, z);
          const float dz = (float)g[u][e] * __builtin_fmaf(sg, t, sg);
          if (MATH >= 2) r[e] = (bf16_t)__builtin_fmaf(k1[e], dz, __builtin_fmaf(k2[e], yv, k3[e]));
          else r[e] = (bf16_t)__builtin_fmaf(k1[e], dz, __builtin_fmaf(sc[e], yv, sh[e]));
        } else {
          r[e] = (bf16_t)((float)g[u][e] + yv);
        }
      }
      st<NT>(o + i * 8, r);
    }
  }
}

typedef void (*kern_t)(const bf16_t*, const bf16_t*, bf16_t*, const float*, long, int);
struct Var { const char* name; kern_t fn; int U; };

__global__ void fill_random(unsigned short* p, long n, unsigned seed) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (unsigned short)(0x3c00u | (h & 0x83ffu));     // bf16 in (-2, 2) away from denormals
  }
}

int main(int argc, char** argv) {
  const bool all = argc > 1 && atoi(argv[1]) == 1;        // 1: every block / grid combination in place as well
  const bool rnd = argc > 2 && atoi(argv[2]) == 1;        // 1: random payload instead of a constant one
  const long sizes[][2] = {{6553600, 32}, {1638400, 64}, {1638400, 32}, {409600, 128}, {409600, 64}, {102400, 256}, {102400, 128}, {25600, 512}, {25600, 256}};
  bf16_t *a, *b, *o; float* k;
  const long maxel = 6553600l * 32;
  CK(hipMalloc(&a, maxel * 2)); CK(hipMalloc(&b, maxel * 2)); CK(hipMalloc(&o, maxel * 2)); CK(hipMalloc(&k, 1280 * 4));
  CK(hipMemset(a, 0x3c, maxel * 2)); CK(hipMemset(b, 0x3c, maxel * 2)); CK(hipMemset(k, 0, 1280 * 4));
  if (rnd) {
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (unsigned short*)a, maxel, 1u);
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (unsigned short*)b, maxel, 2u);
    CK(hipDeviceSynchronize());
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#define V(U, ADJ, NT, MATH) {"U" #U " adj" #ADJ " nt" #NT " math" #MATH, apply2r1w<U, ADJ, NT, MATH>, U}
  const Var vars4[] = {V(1, 0, 1, 4), V(1, 0, 3, 4), V(1, 0, 0, 4), V(2, 1, 1, 4), V(2, 1, 3, 4), V(4, 1, 1, 4), V(4, 1, 3, 4), V(8, 1, 1, 4)};
  const bool fwd = argc > 3 && atoi(argv[3]) == 1;        // 1: the 1-read + 1-write forward pass instead
  const Var vars3[] = {V(1, 0, 1, 2), V(1, 0, 1, 3), V(1, 0, 0, 3), V(1, 0, 3, 3), V(2, 1, 1, 2), V(2, 1, 1, 3), V(4, 1, 1, 2), V(4, 1, 1, 3), V(1, 0, 1, 0), V(4, 1, 1, 0)};
  const Var* vars = fwd ? vars4 : vars3; const int nvars = fwd ? (int)(sizeof(vars4) / sizeof(Var)) : (int)(sizeof(vars3) / sizeof(Var));
  const int blocks[] = {256, 512, 1024};
  const int caps[] = {4096, 16384, 1 << 30};
  for (auto& sz : sizes) {
    const long n16 = sz[0] * sz[1] / 8; const int CC = (int)sz[1] / 8;
    for (int inplace = 0; inplace < 2; ++inplace)
      for (int vi = 0; vi < nvars; ++vi) {
        const Var& v = vars[vi];
        for (int bs : blocks)
          for (int cap : caps) {
            if (inplace && !all && (bs != 256 || cap != 16384)) continue;
            if (bs != 256 && cap == 4096) continue;
            long nb = (n16 + (long)bs * v.U - 1) / ((long)bs * v.U);
            const int grid = (int)(nb < cap ? nb : cap);
            bf16_t* dst = inplace ? b : o;
            // successive launches walk through distinct slices of the 419 MB allocations, so that a tensor smaller than the
            // Infinity Cache is as cold as it is inside a training step
            const long el = sz[0] * sz[1]; const int nrot = (int)(maxel / el < 8 ? maxel / el : 8);
            for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(v.fn, dim3(grid), dim3(bs), 0, 0, a, b, dst, k, n16, CC);
            CK(hipEventRecord(e0, 0));
            const int reps = 16;
            for (int r = 0; r < reps; ++r) {
              const long off = (long)(r % nrot) * el;
              hipLaunchKernelGGL(v.fn, dim3(grid), dim3(bs), 0, 0, a + off, b + off, dst + off, k, n16, CC);
            }
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / reps;
            printf("[%8ld x %3ld] %-22s %s block %4d grid %7d  %8.1f us %6.0f GB/s\n", sz[0], sz[1], v.name, inplace ? "inplace" : "3rd buf", bs, grid, us,
                   (fwd ? 4.0 : 6.0) * sz[0] * sz[1] / us / 1e3);
            fflush(stdout);
          }
      }
  }
  return 0;
}
