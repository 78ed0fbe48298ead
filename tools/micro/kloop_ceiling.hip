// Ceilings of a 256 x 128 x 32 K step on one CU (8 waves, 64 x 64 wave tiles, bf16 MFMA 32x32x16), without any global
// memory: (0) MFMAs from registers, (1) + the fragment reads from LDS, (2) + one s_barrier per K step, (3) = (2) with
// 128 x 64 wave tiles (4 waves).  Diagnostic only (tools/micro/README): hipcc --offload-arch=gfx950 -O3 -o kloop_ceiling kloop_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// MODE 3 / 4: + the real kernel's LDS-DMA staging (3-stage ring, counted vmcnt): every wave issues its share of the 24 one-KB
// pieces of a K step (16 pixel rows x 64 B per piece, row stride 2 * KDIM bytes); 3 = every block re-reads one 256-row tile
// (L2-resident), 4 = every block streams its own rows of a big matrix (HBM).
constexpr int KDIM = 512;
template <int TM, int TN, int MODE>
__global__ __launch_bounds__(64 * (256 / (TM * 32)) * (128 / (TN * 32)), 2) void kloop(const uint32_t* seed, float* out, int steps, const __bf16* amat = nullptr, const __bf16* bmat = nullptr, long arows = 0) {
  constexpr int WAVES_N = 128 / (TN * 32), NT = 64 * (256 / (TM * 32)) * WAVES_N;
  __shared__ __attribute__((aligned(16))) __bf16 lds[(256 + 128) * 32 * 3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  for (int i = tid; i < (256 + 128) * 32 * 3 / 2; i += NT) reinterpret_cast<uint32_t*>(lds)[i] = 0x3c003c00u ^ ((seed[i & 1023] & 0x00ff00ffu));
  __syncthreads();
  f32x16 acc[TN][TM];
  for (int i = 0; i < TN; ++i) for (int j = 0; j < TM; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int fr = lane & 31, fh = lane >> 5;
  bf16x8 wf[TN], xf[TM];
  for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(lds + (256 + wn * TN * 32 + i * 32 + fr) * 32 + fh * 8);
  for (int j = 0; j < TM; ++j) xf[j] = *reinterpret_cast<const bf16x8*>(lds + (wm * TM * 32 + j * 32 + fr) * 32 + fh * 8);
  constexpr int NW = NT / 64, A_PW = 16 / NW, B_PW = 8 / NW;
  __amdgpu_buffer_rsrc_t ra, rb;
  uint32_t avo[A_PW], bvo[B_PW];
  long tile = blockIdx.x;
  if (MODE == 6 || MODE == 7) tile = (blockIdx.x & 7) + (blockIdx.x >> 4) * 8;      // blocks b and b + 8 (same XCD) stream the same rows
  auto dma = [&](int kt, int buf) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (MODE >= 3) {
      char* As = reinterpret_cast<char*>(lds + buf * (256 + 128) * 32);
      char* Bs = As + 256 * 64;
      const int kk = MODE == 5 ? ((kt % (KDIM / 32)) >> 1) * 2 : kt % (KDIM / 32);
      uint32_t base = 0;
      if (MODE == 5) { const long t = (tile + (long)(kt / (KDIM / 32)) * gridDim.x) % (arows / 256); base = (uint32_t)(t * 256 * KDIM * 2 + ((kt & 1) ? 128 * KDIM * 2 : 0)) ; }
      if (MODE == 7) { const long t = (tile + (long)(kt / (KDIM / 32)) * gridDim.x) % 100; base = (uint32_t)(t * 256 * KDIM * 2); }      // 26 MB: Infinity-Cache-resident
      if (MODE == 4 || MODE == 6) { const long t = (tile + (long)(kt / (KDIM / 32)) * gridDim.x) % (arows / 256); base = (uint32_t)(t * 256 * KDIM * 2); }
#pragma unroll
      for (int i = 0; i < A_PW; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(As + (wave * A_PW + i) * 1024), 16, avo[i] + base, kk * 64, 0, 0);
#pragma unroll
      for (int i = 0; i < B_PW; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(Bs + (wave * B_PW + i) * 1024), 16, bvo[i], kk * 64, 0, 0);
    }
#endif
  };
  if constexpr (MODE >= 3) {
#if defined(__HIP_DEVICE_COMPILE__)
    ra = __builtin_amdgcn_make_buffer_rsrc((void*)amat, 0, 0xFFFFFFF0u, 0x00020000);
    rb = __builtin_amdgcn_make_buffer_rsrc((void*)bmat, 0, 0xFFFFFFF0u, 0x00020000);
#endif
    for (int i = 0; i < A_PW; ++i) {
      const int row = (wave * A_PW + i) * 16 + (lane >> 2); avo[i] = (uint32_t)((row * KDIM + ((lane & 3) ^ ((row >> 2) & 3)) * 8) * 2);
      if (MODE == 5) { const int r8 = (wave * A_PW + i) * 8 + (lane >> 3); avo[i] = (uint32_t)((r8 * KDIM + (lane & 7) * 8) * 2); }   // 8 rows x 128 B: whole lines
    }
    for (int i = 0; i < B_PW; ++i) { const int row = (wave * B_PW + i) * 16 + (lane >> 2); bvo[i] = (uint32_t)((row * KDIM + ((lane & 3) ^ ((row >> 2) & 3)) * 8) * 2); }
    dma(0, 0); dma(1, 1);
  }
  for (int kt = 0; kt < steps; ++kt) {
    const __bf16* As = lds + (kt % 3) * (256 + 128) * 32;
    const __bf16* Bs = As + 256 * 32;
    if constexpr (MODE >= 3) wait_vm<A_PW + B_PW>();
    if (MODE >= 2) __builtin_amdgcn_s_barrier();
    if constexpr (MODE >= 3) dma(kt + 2, (kt + 2) % 3);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (MODE >= 1) {
#pragma unroll
        for (int i = 0; i < TN; ++i) { int row = wn * TN * 32 + i * 32 + fr; wf[i] = *reinterpret_cast<const bf16x8*>(Bs + row * 32 + ((ks * 2 + fh) ^ ((row >> 2) & 3)) * 8); }
#pragma unroll
        for (int j = 0; j < TM; ++j) { int row = wm * TM * 32 + j * 32 + fr; xf[j] = *reinterpret_cast<const bf16x8*>(As + row * 32 + ((ks * 2 + fh) ^ ((row >> 2) & 3)) * 8); }
      } else {
        for (int i = 0; i < TN; ++i) asm volatile("" : "+v"(wf[i]));
      }
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < TN; ++i) for (int j = 0; j < TM; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  if (s == 12345.678f) out[blockIdx.x * NT + tid] = s;
}

static __bf16 *g_a, *g_b; static long g_rows;
template <int TM, int TN, int MODE>
static void run(const char* name, int blocks, const uint32_t* seed, float* out) {
  constexpr int NT = 64 * (256 / (TM * 32)) * (128 / (TN * 32));
  const int steps = 2048;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((kloop<TM, TN, MODE>), dim3(blocks), dim3(NT), 0, 0, seed, out, steps, g_a, g_b, g_rows);
  hipEventRecord(e0);
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((kloop<TM, TN, MODE>), dim3(blocks), dim3(NT), 0, 0, seed, out, steps, g_a, g_b, g_rows);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double flop = 2.0 * 256 * 128 * 32 * steps * blocks;
  printf("%-44s blocks %4d  %8.1f us  %7.1f TF/s  %6.3f us per K step per block-slot\n", name, blocks, ms * 1e3, flop / ms / 1e9, ms * 1e3 / steps / ((blocks + 255) / 256));
}

int main() {
  uint32_t h[1024]; for (int i = 0; i < 1024; ++i) h[i] = rand();
  uint32_t* seed; float* out; hipMalloc(&seed, 4096); hipMalloc(&out, 1 << 24); hipMemcpy(seed, h, 4096, hipMemcpyHostToDevice);
  g_rows = 1 << 20;                                  // 1 Mi rows x 512 x 2 B = 1 GiB
  hipMalloc(&g_a, (size_t)g_rows * KDIM * 2 + 4096); hipMalloc(&g_b, 128 * KDIM * 2 + 4096);
  hipMemset(g_a, 0x3c, (size_t)g_rows * KDIM * 2); hipMemset(g_b, 0x3c, 128 * KDIM * 2);
  for (int blocks : {200, 256, 512}) {
    run<2, 2, 0>("8 waves 64x64: MFMA from registers", blocks, seed, out);
    run<2, 2, 1>("8 waves 64x64: + LDS fragment reads", blocks, seed, out);
    run<2, 2, 2>("8 waves 64x64: + s_barrier per K step", blocks, seed, out);
    run<4, 2, 0>("4 waves 128x64: MFMA from registers", blocks, seed, out);
    run<4, 2, 1>("4 waves 128x64: + LDS fragment reads", blocks, seed, out);
    run<4, 2, 2>("4 waves 128x64: + s_barrier per K step", blocks, seed, out);
    run<2, 2, 3>("8 waves 64x64: + LDS-DMA ring, L2-resident", blocks, seed, out);
    run<2, 2, 4>("8 waves 64x64: + LDS-DMA ring, HBM stream", blocks, seed, out);
    run<2, 2, 5>("8 waves 64x64: HBM stream, whole-line pieces", blocks, seed, out);
    run<2, 2, 6>("8 waves 64x64: HBM stream shared by 2 blocks/XCD", blocks, seed, out);
    run<2, 2, 7>("8 waves 64x64: 26 MB stream (MALL), shared by 2", blocks, seed, out);
    run<4, 2, 3>("4 waves 128x64: + LDS-DMA ring, L2-resident", blocks, seed, out);
    run<4, 2, 4>("4 waves 128x64: + LDS-DMA ring, HBM stream", blocks, seed, out);
  }
  return 0;
}
