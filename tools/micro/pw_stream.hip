// A shallow pointwise convolution as a STREAM: can y[M][COUT] = x[M][CIN] . w[COUT][CIN]^T (bf16 in, fp32 accumulate, bf16 out,
// + per-channel sum / sum of squares as the forward conv's BatchNorm statistics) run at the apply passes' bandwidth when every
// wave works alone - pixel fragments straight from global memory into the MFMA operand registers (no LDS, no barriers),
// weights resident in registers, output channels permuted over the MFMA rows so that a lane ends up with 16 consecutive
// channels of one pixel (two 16-byte stores)?   The library's LDS-DMA kernel runs these layers at 4.5 - 4.9 TB/s.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/pw_stream tools/micro/pw_stream.hip && tools/micro/pw_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
typedef __bf16 bf16_t;
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// MFMA 32x32x16: A = weights (row rho <-> out channel), B = pixels (column n <-> pixel).  Lane l: n / rho = l & 31, k half h = l >> 5.
// D[e] of lane l: row rho = 8 (e >> 2) + 4 h + (e & 3), column n.  Row rho = 8 a + 4 h + b carries channel 16 h + 4 a + b, so lane
// (n, h) holds channels 16 h .. 16 h + 15 of pixel n in e order.
template <int CIN, int COUT, int U>
__global__ __launch_bounds__(256) void pw_fwd(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, bf16_t* __restrict__ y,
                                               float* __restrict__ part, long M) {
  constexpr int KC = CIN / 16, NT = COUT / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, h = lane >> 5;
  bf16x8 wf[NT][KC];
  {
    const int a = n >> 3, hh = (n >> 2) & 1, b = n & 3;
    const int ch = 16 * hh + 4 * a + b;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int k = 0; k < KC; ++k) wf[t][k] = *reinterpret_cast<const bf16x8*>(w + (long)(t * 32 + ch) * CIN + k * 16 + h * 8);
  }
  float s0[NT][16], s1[NT][16];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) s0[t][e] = s1[t][e] = 0.f;
  const long wave_px = 32l * U;                                   // pixels per wave per iteration
  const long step = (long)gridDim.x * 4 * wave_px;
  for (long p0 = ((long)blockIdx.x * 4 + wave) * wave_px; p0 < M; p0 += step) {
    bf16x8 xf[U][KC];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long p = p0 + u * 32 + n;
#pragma unroll
      for (int k = 0; k < KC; ++k)
        xf[u][k] = p < M ? __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(x + p * CIN + k * 16 + h * 8)) : bf16x8{};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long p = p0 + u * 32 + n;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int k = 0; k < KC; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[t][k], xf[u][k], acc, 0, 0, 0);
        bf16x8 o0, o1;
#pragma unroll
        for (int e = 0; e < 8; ++e) { o0[e] = (bf16_t)acc[e]; o1[e] = (bf16_t)acc[8 + e]; }
        if (p < M) {
          *reinterpret_cast<bf16x8*>(y + p * COUT + t * 32 + 16 * h) = o0;
          *reinterpret_cast<bf16x8*>(y + p * COUT + t * 32 + 16 * h + 8) = o1;
#pragma unroll
          for (int e = 0; e < 16; ++e) { s0[t][e] += acc[e]; s1[t][e] = __builtin_fmaf(acc[e], acc[e], s1[t][e]); }
        }
      }
    }
  }
  // statistics: over the 32 pixels of a half-wave (xor 1 .. 16), then the block's four waves through LDS -> part[block][2][COUT]
  __shared__ float red[4][2][COUT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float a = s0[t][e], b = s1[t][e];
#pragma unroll
      for (int o = 1; o < 32; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
      if (n == 0) { red[wave][0][t * 32 + 16 * h + e] = a; red[wave][1][t * 32 + 16 * h + e] = b; }
    }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * COUT; i += 256) {
    const int m = i / COUT, c = i % COUT;
    part[(long)blockIdx.x * 2 * COUT + i] = red[0][m][c] + red[1][m][c] + red[2][m][c] + red[3][m][c];
  }
}

static float bf(unsigned short v) { unsigned u = (unsigned)v << 16; float f; memcpy(&f, &u, 4); return f; }
static unsigned short tobf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }

template <int CIN, int COUT, int U>
void run(long M, int cap, bool check) {
  std::vector<unsigned short> hx((size_t)M * CIN), hw((size_t)COUT * CIN);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  if (check) for (auto& v : hx) v = tobf(rnd() * 2);
  for (auto& v : hw) v = tobf(rnd() * 0.5f);
  bf16_t *x, *w, *y; float* part;
  CK(hipMalloc(&x, (size_t)M * CIN * 2)); CK(hipMalloc(&w, (size_t)COUT * CIN * 2)); CK(hipMalloc(&y, (size_t)M * COUT * 2));
  if (check) CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); else CK(hipMemset(x, 0x3c, (size_t)M * CIN * 2));
  CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  long nb = (M + 128l * U - 1) / (128l * U);
  const int grid = (int)(nb < cap ? nb : cap);
  CK(hipMalloc(&part, (size_t)grid * 2 * COUT * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((pw_fwd<CIN, COUT, U>), dim3(grid), dim3(256), 0, 0, x, w, y, part, M);
  CK(hipEventRecord(e0, 0));
  const int reps = check ? 1 : 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((pw_fwd<CIN, COUT, U>), dim3(grid), dim3(256), 0, 0, x, w, y, part, M);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps;
  printf("%4d -> %4d  M %8ld  U%d grid %6d  %8.1f us  %6.0f GB/s", CIN, COUT, M, U, grid, us, 2.0 * M * (CIN + COUT) / us / 1e3);
  if (check) {
    std::vector<unsigned short> hy((size_t)M * COUT); std::vector<float> hp((size_t)grid * 2 * COUT);
    CK(hipMemcpy(hy.data(), y, hy.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hp.data(), part, hp.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, ssum = 0, sref = 0;
    for (long p = 0; p < M; ++p)
      for (int c = 0; c < COUT; ++c) {
        float r = 0;
        for (int k = 0; k < CIN; ++k) r += bf(hx[p * CIN + k]) * bf(hw[(size_t)c * CIN + k]);
        worst = fmax(worst, fabs(bf(hy[p * COUT + c]) - r) / (1.0 + fabs(r)));
        if (c == 5) sref += r;
      }
    for (int b = 0; b < grid; ++b) ssum += hp[(size_t)b * 2 * COUT + 5];
    printf("   max rel err %.4f, channel-5 sum %.3f vs %.3f", worst, ssum, sref);
  }
  printf("\n"); fflush(stdout);
  CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(y)); CK(hipFree(part));
}

int main() {
  run<64, 32, 1>(3000, 1 << 30, true);
  run<32, 32, 2>(2777, 1 << 30, true);
  run<64, 64, 2>(1111, 7, true);
  const long M = 1638400;
  for (int cap : {1 << 30, 16384, 4096, 2048}) {
    run<64, 32, 1>(M, cap, false); run<64, 32, 2>(M, cap, false); run<64, 32, 4>(M, cap, false);
    run<32, 32, 1>(M, cap, false); run<32, 32, 2>(M, cap, false); run<32, 32, 4>(M, cap, false);
    run<64, 64, 1>(M, cap, false); run<64, 64, 2>(M, cap, false);
    run<32, 64, 2>(M, cap, false);
  }
  run<128, 64, 1>(409600, 1 << 30, false); run<128, 64, 2>(409600, 1 << 30, false); run<64, 64, 2>(409600, 1 << 30, false); run<128, 128, 1>(409600, 1 << 30, false);
  return 0;
}
