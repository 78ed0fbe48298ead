// Weight gradient of a convolution (aten::convolution_backward dW) for gfx950.
//
//   dW[n][k] = sum_m dY[m][n] * im2col(X)[m][k],  m = (b, oy, ox), k = (kh, kw, ci)
//
// Both MFMA operands are indexed [reduction m][n or k] in memory (channels-last), i.e. the reduction
// index is the slow one, so the fragments are read from LDS with gfx950's transposing
// ds_read_b64_tr_b16 (row stride = 64 or 192 mod 256 bytes => conflict-free).  The huge reduction
// (m up to 6.5M rows) is split across blocks; each split writes an fp32 partial slab, and a second
// kernel sums the slabs in a fixed order (deterministic, no float atomics) while un-permuting k back
// to torch's [Cout][Cin][KH][KW] fp32 layout.
#include <stdlib.h>

#include "kodhip_common.h"

namespace {

struct WgradArgs {
  const bf16_t* x;
  const bf16_t* dy;
  float* part;
  int B, Hs, Ws, ldx, xcoff, Cin;
  int Ho, Wo, M;
  int N, K, Kp;
  int KH, KW, SH, SW, PH, PW;
  int ldy, ycoff;
  int m_per_split, splits;
  int tiles_n, tiles_k;
  uint32_t magic_cin, magic_kw;
  uint32_t magic_hwo, magic_wo;      // ceil(2^32 / d); 0 encodes d == 1
  // dual form (kodhip_conv_wgrad_dual): two layers with the same input, N = 2 * n_half slab rows; n tiles of the second
  // half take their dY from dy2.  n_half = 0: one layer
  const bf16_t* dy2;
  int n_half;
  int linear_map;                    // block -> (tile, split) without the split-per-XCD banding (wg_map)
};

// block -> (tile, split).  Default: the blocks of one split sit on ONE XCD (bid & 7), so the input rows a split streams are
// fetched from HBM once for all of its n / k tiles.  That banding idles XCDs when the split count is small and not a multiple
// of 8 - yv5m's 384 -> 768 stride-2 layer has 162 tiles and 3 splits: five of eight XCDs had nothing to do (555 us; 333 with six
// splits) - so layers with few splits (deep layers: their operands live in the caches anyway) take a linear map, which deals
// every split's tiles round-robin over the XCDs.  The slabs are indexed by split either way: results do not depend on the map.
__device__ __forceinline__ bool wg_map(const WgradArgs& a, int bid, int tiles, int& tile, int& split) {
  if (a.linear_map) {
    if (bid >= tiles * a.splits) return false;
    split = bid % a.splits;
    tile = bid / a.splits;
    return true;
  }
  const int j = bid >> 3;
  tile = j % tiles;
  split = (j / tiles) * 8 + (bid & 7);
  return split < a.splits;
}
__host__ inline int wg_grid(WgradArgs& a) {
  const int tiles = a.tiles_n * a.tiles_k;
  static int mode = -1;               // KODHIP_WGRAD_LINEAR=0: A/B knob
  if (mode < 0) { const char* e = getenv("KODHIP_WGRAD_LINEAR"); mode = e ? atoi(e) : 1; }
  a.linear_map = (mode && a.splits % 8 != 0 && a.splits < 64) ? 1 : 0;
  return a.linear_map ? tiles * a.splits : cdiv(a.splits, 8) * 8 * tiles;
}

__host__ __device__ constexpr int row_bytes(int T) { return (T * 2) % 128 == 0 ? T * 2 + 64 : T * 2; }

// Block tile = (WN*RN*32) channels x (WK*RK*32) k, 64*WN*WK threads, 32 reduction rows per step.
template <int WN, int WK, int RN, int RK>
__global__ __launch_bounds__(64 * WN * WK) void conv_wgrad_kernel(WgradArgs a) {
  constexpr int NT = 64 * WN * WK;
  constexpr int TNB = WN * RN * 32;
  constexpr int TKB = WK * RK * 32;
  constexpr int SY = row_bytes(TNB);
  constexpr int SX = row_bytes(TKB);
  constexpr int YC = TNB / 8, XC = TKB / 8;
  constexpr int YL = (32 * YC + NT - 1) / NT;    // chunks per thread
  constexpr int XL = (32 * XC + NT - 1) / NT;
  constexpr int STAGE = 32 * SY + 32 * SX;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wn = wave / WK, wk = wave % WK;

  const int bid = blockIdx.x;
  const int tiles = a.tiles_n * a.tiles_k;
  int tile, split;
  if (!wg_map(a, bid, tiles, tile, split)) return;
  const int n0 = (tile % a.tiles_n) * TNB;
  const int k0 = (tile / a.tiles_n) * TKB;
  const int m_begin = split * a.m_per_split;
  int m_end = m_begin + a.m_per_split;
  if (m_end > a.M) m_end = a.M;

  // fixed per-thread k-chunk decomposition for the X gather
  const int xkc = tid % XC;                 // NT % XC == 0 for all instantiations
  const int xrow0 = tid / XC;
  constexpr int XRS = NT / XC;              // row step between a thread's chunks
  const int xk = k0 + xkc * 8;
  const uint32_t tap = __umulhi((uint32_t)xk, a.magic_cin);
  const int xci = xk - (int)tap * a.Cin;
  const int xkh = (a.KW == 1) ? (int)tap : (int)__umulhi(tap, a.magic_kw);
  const int xkw = (int)tap - xkh * a.KW;
  const bool xkvalid = xk < a.K;

  const int ykc = tid % YC;
  const int yrow0 = tid / YC;
  constexpr int YRS = NT / YC;
  const bool ynvalid = (n0 + ykc * 8) < a.N;

  const int HWo = a.Ho * a.Wo;

  u32x4 xreg[XL], yreg[YL];
  // incremental (b, oy, ox) of each gather row: one division at the start, then +32 rows per step
  int sb[XL], soy[XL], sox[XL];
#pragma unroll
  for (int i = 0; i < XL; ++i) {
    int m = m_begin + xrow0 + i * XRS;
    int b = m / HWo;
    int rem = m - b * HWo;
    soy[i] = rem / a.Wo;
    sox[i] = rem - soy[i] * a.Wo;
    sb[i] = b;
  }
  auto load_tile = [&](int mb) {
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      int row = xrow0 + i * XRS;
      int m = mb + row;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (row < 32 && m < m_end && xkvalid) {
        int iy = soy[i] * a.SH - a.PH + xkh;
        int ix = sox[i] * a.SW - a.PW + xkw;
        if (iy >= 0 && iy < a.Hs && ix >= 0 && ix < a.Ws)
          v = *reinterpret_cast<const u32x4*>(a.x + (size_t)((sb[i] * a.Hs + iy) * a.Ws + ix) * a.ldx + a.xcoff + xci);
      }
      xreg[i] = v;
      sox[i] += 32;
      while (sox[i] >= a.Wo) {
        sox[i] -= a.Wo;
        if (++soy[i] >= a.Ho) { soy[i] = 0; ++sb[i]; }
      }
    }
#pragma unroll
    for (int i = 0; i < YL; ++i) {
      int row = yrow0 + i * YRS;
      int m = mb + row;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (row < 32 && m < m_end && ynvalid)
        v = *reinterpret_cast<const u32x4*>(a.dy + (size_t)m * a.ldy + a.ycoff + n0 + ykc * 8);
      yreg[i] = v;
    }
  };
  auto store_tile = [&](int buf) {
    unsigned char* Ys = lds + buf * STAGE;
    unsigned char* Xs = Ys + 32 * SY;
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      int row = xrow0 + i * XRS;
      if (row < 32) *reinterpret_cast<u32x4*>(Xs + row * SX + xkc * 16) = xreg[i];
    }
#pragma unroll
    for (int i = 0; i < YL; ++i) {
      int row = yrow0 + i * YRS;
      if (row < 32) *reinterpret_cast<u32x4*>(Ys + row * SY + ykc * 16) = yreg[i];
    }
  };

  f32x16 acc[RN][RK];
#pragma unroll
  for (int i = 0; i < RN; ++i)
#pragma unroll
    for (int jj = 0; jj < RK; ++jj)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][jj][e] = 0.f;

  // transposed-read lane geometry (see file header): rows = reduction index, columns = n or k
  const int tr_row = 8 * (lane >> 5) + ((lane & 15) >> 2);
  const int tr_col = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

  const int nsteps = (m_end - m_begin + 31) / 32;
  if (nsteps > 0) {
    load_tile(m_begin);
    store_tile(0);
  }
  __syncthreads();
  for (int st = 0; st < nsteps; ++st) {
    const int buf = st & 1;
    if (st + 1 < nsteps) load_tile(m_begin + (st + 1) * 32);
    const unsigned char* Ys = lds + buf * STAGE;
    const unsigned char* Xs = Ys + 32 * SY;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 yf[RN], xf[RK];
#pragma unroll
      for (int i = 0; i < RN; ++i) {
        const unsigned char* p = Ys + (ks * 16 + tr_row) * SY + ((wn * RN + i) * 32 + tr_col) * 2;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 4 * SY));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        s16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        yf[i] = __builtin_bit_cast(bf16x8, t);
      }
#pragma unroll
      for (int jj = 0; jj < RK; ++jj) {
        const unsigned char* p = Xs + (ks * 16 + tr_row) * SX + ((wk * RK + jj) * 32 + tr_col) * 2;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 4 * SX));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        s16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        xf[jj] = __builtin_bit_cast(bf16x8, t);
      }
#pragma unroll
      for (int i = 0; i < RN; ++i)
#pragma unroll
        for (int jj = 0; jj < RK; ++jj)
          acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yf[i], xf[jj], acc[i][jj], 0, 0, 0);
    }
    if (st + 1 < nsteps) store_tile(buf ^ 1);
    __syncthreads();
  }

  // D[n][k]: k = lane & 31, n = 8*(e>>2) + 4*(lane>>5) + (e&3)
  float* slab = a.part + (size_t)split * a.N * a.Kp;
#pragma unroll
  for (int i = 0; i < RN; ++i)
#pragma unroll
    for (int jj = 0; jj < RK; ++jj) {
      int k = k0 + (wk * RK + jj) * 32 + (lane & 31);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        int n = n0 + (wn * RN + i) * 32 + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3);
        if (n < a.N && k < a.Kp) slab[(size_t)n * a.Kp + k] = acc[i][jj][e];
      }
    }
}

// ---- LDS-DMA variant --------------------------------------------------------------------------------------------
// Same tiling and fragment reads, but the 32-row operand tiles go HBM -> LDS with buffer_load..lds (no VGPR staging)
// through an NST-deep ring.  The loop is fill-latency-bound: a CU streams (bytes in flight) / latency, and two
// resident blocks x one 16 KB register-staged tile (the kernel above) cover only a third of what the MFMAs can eat;
// here NST-1 tiles per block are in flight.  A DMA instruction writes 64 lanes x 16 B linearly, so LDS rows are
// unpadded; the transposed fragment reads stay conflict-free through an XOR of the 16-byte chunk index with the row
// (applied on the source side: LDS slot (row, c') holds chunk c = c' ^ swz(row)).  Padding / ragged rows / ragged
// channels are out-of-range buffer offsets, which the buffer unit zero-fills.
__host__ __device__ constexpr int swz_mask(int row_b) { return row_b % 256 == 0 ? 3 : (row_b % 128 == 0 ? 2 : 0); }
__device__ __forceinline__ int swz(int row, int mask) { return mask == 3 ? 4 * (row & 3) : (mask == 2 ? 4 * ((row >> 1) & 1) : 0); }

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// GM = how the X gather address of a step is produced: 1 pointwise (1x1 / stride 1 / no padding: a plain strided
// stream like dY, no address arithmetic in the loop), 0 two magic divisions per step (any geometry).  A compile-time
// choice: the loop is instruction-issue-bound (rocprofv3: MFMA pipe utilisation 12 %), a run-time switch would keep
// both address generators and their branches in it.  (Measured and dropped: carrying (b, oy, ox) per lane with
// branch-free wraps costs as many VALU instructions as the divisions.)
template <int WN, int WK, int RN, int RK, int NST, int RS, int GM>
__global__ __launch_bounds__(64 * WN * WK) void conv_wgrad_dma_kernel(WgradArgs a, uint32_t x_bytes, uint32_t dy_bytes) {
  constexpr int NW = WN * WK;
  constexpr int TNB = WN * RN * 32;
  constexpr int TKB = WK * RK * 32;
  constexpr int RBY = TNB * 2, RBX = TKB * 2;             // unpadded row bytes
  constexpr int NIY = RS * RBY / 1024, NIX = RS * RBX / 1024;   // DMA instructions per tile (RS reduction rows)
  constexpr int KS = RS / 16;                                   // MFMA k-groups per tile
  constexpr int NI = NIY + NIX;
  constexpr int SLOTS = (NI + NW - 1) / NW;               // per wave
  constexpr int STAGE = RS * RBY + RS * RBX;
  constexpr int MY = swz_mask(RBY), MX = swz_mask(RBX);
  static_assert((RS * RBY) % 1024 == 0 && (RS * RBX) % 1024 == 0, "tile rows must fill whole DMA instructions");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NST * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave / WK, wk = wave % WK;

  const int bid = blockIdx.x;
  const int tiles = a.tiles_n * a.tiles_k;
  int tile, split;
  if (!wg_map(a, bid, tiles, tile, split)) return;
  // slab rows n0 .. of this block; dY columns ny0 .. of its layer (dual form: the second half of the n tiles reads dy2)
  const int nt = tile % a.tiles_n;
  const bool half2 = a.n_half && nt >= (a.tiles_n >> 1);
  const int n0 = half2 ? a.n_half + (nt - (a.tiles_n >> 1)) * TNB : nt * TNB;
  const int ny0 = half2 ? n0 - a.n_half : n0;
  const int ny_hi = a.n_half ? a.n_half : a.N;                       // dY columns of one layer
  const int n_hi = (a.n_half && !half2) ? a.n_half : a.N;            // slab rows this block may write
  const int k0 = (tile / a.tiles_n) * TKB;
  const int m_begin = split * a.m_per_split;
  int m_end = m_begin + a.m_per_split;
  if (m_end > a.M) m_end = a.M;
  const int HWo = a.Ho * a.Wo;

#if defined(__HIP_DEVICE_COMPILE__)
  __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)(half2 ? a.dy2 : a.dy), 0, dy_bytes, 0x00020000);
#endif

  // ---- per-lane constants of this wave's DMA slots: instruction t = wave + i*NW; t < NIY feeds the dY tile, else X.
  //      The gather address of a step is recomputed from the row index with two exact magic divisions (branch-free,
  //      ~30 VALU per slot): the loop is instruction-issue-bound on address generation, not on MFMA or bandwidth.
  int s_row[SLOTS];                 // tile row this lane fills
  uint32_t s_off[SLOTS];            // dY (and X of a pointwise conv): running byte offset; X: byte offset of (xcoff + ci)
  bool s_ok[SLOTS];                 // channel / k column in range
  int s_cy[SLOTS], s_cx[SLOTS];     // X: kh - PH, kw - PW
  int my_slots = 0;
  constexpr bool pointwise = GM == 1;
  // slot roles: instruction t = wave + i * NW feeds dY when t < NIY, else X.  When NW divides NIY and NI the role of
  // slot i is the same for every wave (compile-time): no per-slot branches in the loop
  constexpr bool ROLE_STATIC = (NIY % NW == 0) && (NI % NW == 0);
#pragma unroll
  for (int i = 0; i < SLOTS; ++i) {
    const int t = wave + i * NW;
    s_row[i] = 0; s_off[i] = 0; s_ok[i] = false; s_cy[i] = s_cx[i] = 0;
    if (!ROLE_STATIC && t >= NI) continue;
    ++my_slots;
    if (ROLE_STATIC ? (i < NIY / NW) : (t < NIY)) {
      const int L = t * 1024 + lane * 16;
      const int row = L / RBY, pc = (L % RBY) >> 4;
      const int c = pc ^ swz(row, MY);
      s_row[i] = row;
      s_ok[i] = (ny0 + c * 8) < ny_hi;
      s_off[i] = (uint32_t)(((long)(m_begin + row) * a.ldy + a.ycoff + ny0 + c * 8) * 2);
    } else {
      const int L = (t - NIY) * 1024 + lane * 16;
      const int row = L / RBX, pc = (L % RBX) >> 4;
      const int c = pc ^ swz(row, MX);
      const int k = k0 + c * 8;
      const uint32_t tap = __umulhi((uint32_t)k, a.magic_cin);
      const int ci = k - (int)tap * a.Cin;
      const int kh = (a.KW == 1) ? (int)tap : (int)__umulhi(tap, a.magic_kw);
      s_row[i] = row;
      s_ok[i] = k < a.K;
      s_cy[i] = kh - a.PH; s_cx[i] = ((int)tap - kh * a.KW) - a.PW;
      s_off[i] = (uint32_t)((a.xcoff + ci) * 2);
      if (pointwise) s_off[i] = (uint32_t)(((long)(m_begin + row) * a.ldx + a.xcoff + ci) * 2);
    }
  }
  const uint32_t magic_hwo = a.magic_hwo, magic_wo = a.magic_wo;
  const uint32_t ldx2 = (uint32_t)a.ldx * 2u, ystep = (uint32_t)(RS * a.ldy * 2), xstep = (uint32_t)(RS * a.ldx * 2);

  int issued = 0;                   // tiles issued so far (tile index = reduction step)
  auto issue = [&](int stage) {
    unsigned char* Ys = lds + stage * STAGE;
    const int mb = m_begin + issued * RS;
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      const int t = wave + i * NW;
      if (!ROLE_STATIC && t >= NI) continue;
      const int m = mb + s_row[i];
      uint32_t vo;
      bool ok = s_ok[i] && m < m_end;
      const bool is_y = ROLE_STATIC ? (i < NIY / NW) : (t < NIY);
      if (is_y) {
        vo = s_off[i];
        s_off[i] += ystep;
      } else if (pointwise) {
        vo = s_off[i];
        s_off[i] += xstep;
      } else {
        // q = m / d with magic = ceil(2^32 / d): the estimate is q or q + 1 for any 31-bit m
        uint32_t b = magic_hwo ? __umulhi((uint32_t)m, magic_hwo) : (uint32_t)m;
        int rem = m - (int)b * HWo;
        if (rem < 0) { rem += HWo; --b; }
        uint32_t oy = magic_wo ? __umulhi((uint32_t)rem, magic_wo) : (uint32_t)rem;
        int ox = rem - (int)oy * a.Wo;
        if (ox < 0) { ox += a.Wo; --oy; }
        const int iy = (int)oy * a.SH + s_cy[i];
        const int ix = ox * a.SW + s_cx[i];
        ok = ok && (unsigned)iy < (unsigned)a.Hs && (unsigned)ix < (unsigned)a.Ws;
        vo = (((uint32_t)b * (uint32_t)a.Hs + (uint32_t)iy) * (uint32_t)a.Ws + (uint32_t)ix) * ldx2 + s_off[i];
      }
      vo = ok ? vo : 0xFFFFFFF0u;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(WG_ABL_NODMA)
      if (is_y)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, (__attribute__((address_space(3))) void*)(Ys + t * 1024), 16, vo, 0, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(Ys + t * 1024), 16, vo, 0, 0, 0);
#else
      (void)vo; (void)Ys;
#endif
    }
    ++issued;
  };

  f32x16 acc[RN][RK];
#pragma unroll
  for (int i = 0; i < RN; ++i)
#pragma unroll
    for (int jj = 0; jj < RK; ++jj)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][jj][e] = 0.f;

  const int tr_row = 8 * (lane >> 5) + ((lane & 15) >> 2);
  const int tr_col = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const int sy = swz(tr_row, MY), sx = swz(tr_row, MX);    // the rows a lane reads differ by multiples of 4 only
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  uint32_t yoff[RN], xoff[RK];
#pragma unroll
  for (int i = 0; i < RN; ++i) {
    const int col = (wn * RN + i) * 32 + tr_col;
    yoff[i] = (uint32_t)(tr_row * RBY + (((col >> 3) ^ sy) << 4) + (col & 7) * 2);
  }
#pragma unroll
  for (int jj = 0; jj < RK; ++jj) {
    const int col = (wk * RK + jj) * 32 + tr_col;
    xoff[jj] = (uint32_t)(tr_row * RBX + (((col >> 3) ^ sx) << 4) + (col & 7) * 2);
  }

  const int nsteps = (m_end - m_begin + RS - 1) / RS;
#pragma unroll
  for (int p = 0; p < NST - 1; ++p)
    if (p < nsteps) issue(p);
  for (int st = 0; st < nsteps; ++st) {
    // tile st must have landed; up to NST-2 newer tiles (my_slots instructions each, SLOTS or SLOTS-1) stay in flight
    const int newer = nsteps - 1 - st;
    if (newer >= NST - 2) {
      if (my_slots == SLOTS) wait_vm<(NST - 2) * SLOTS>(); else wait_vm<(NST - 2) * (SLOTS - 1)>();
    } else if (newer == 1 && NST > 3) {
      if (my_slots == SLOTS) wait_vm<SLOTS>(); else wait_vm<SLOTS - 1>();
    } else {
      wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();
#ifndef WG_ABL_NOISSUE
    if (st + NST - 1 < nsteps) issue((st + NST - 1) % NST);
#endif
    // Fragment reads are issued as inline asm: the compiler cannot tell which ring stage a ds_read touches and
    // would otherwise drain every outstanding LDS-DMA (s_waitcnt vmcnt(0)) in front of them, which serialises the
    // ring.  Both k-halves are read up front; the first MFMA group waits for its half only (LDS returns in order).
    const uint32_t ys = lds_base + (uint32_t)((st % NST) * STAGE);
    const uint32_t xs = ys + RS * RBY;
    s16x4 ylo[KS][RN], yhi[KS][RN], xlo[KS][RK], xhi[KS][RK];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int i = 0; i < RN; ++i) {
        const uint32_t p = ys + yoff[i] + (uint32_t)(ks * 16 * RBY);
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(ylo[ks][i]) : "v"(p) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(yhi[ks][i]) : "v"(p + 4 * RBY) : "memory");
      }
#pragma unroll
      for (int jj = 0; jj < RK; ++jj) {
        const uint32_t p = xs + xoff[jj] + (uint32_t)(ks * 16 * RBX);
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(xlo[ks][jj]) : "v"(p) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(xhi[ks][jj]) : "v"(p + 4 * RBX) : "memory");
      }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      // wait for this k-group's reads (LDS returns in order; the counter saturates at 15); the registers are
      // operands of the empty asm below so that the MFMAs cannot be hoisted above the wait
      constexpr int PER = 2 * (RN + RK);
      if (ks == KS - 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      else if ((KS - 1 - ks) * PER >= 15) asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
      else if (ks == KS - 2) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(PER < 15 ? PER : 15) : "memory");
      else asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * PER < 15 ? 2 * PER : 15) : "memory");
#pragma unroll
      for (int i = 0; i < RN; ++i) asm volatile("" : "+v"(ylo[ks][i]), "+v"(yhi[ks][i]));
#pragma unroll
      for (int jj = 0; jj < RK; ++jj) asm volatile("" : "+v"(xlo[ks][jj]), "+v"(xhi[ks][jj]));
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      bf16x8 yf[RN], xf[RK];
#pragma unroll
      for (int i = 0; i < RN; ++i) {
        s16x8 t = {ylo[ks][i][0], ylo[ks][i][1], ylo[ks][i][2], ylo[ks][i][3], yhi[ks][i][0], yhi[ks][i][1], yhi[ks][i][2], yhi[ks][i][3]};
        yf[i] = __builtin_bit_cast(bf16x8, t);
      }
#pragma unroll
      for (int jj = 0; jj < RK; ++jj) {
        s16x8 t = {xlo[ks][jj][0], xlo[ks][jj][1], xlo[ks][jj][2], xlo[ks][jj][3], xhi[ks][jj][0], xhi[ks][jj][1], xhi[ks][jj][2], xhi[ks][jj][3]};
        xf[jj] = __builtin_bit_cast(bf16x8, t);
      }
#pragma unroll
      for (int i = 0; i < RN; ++i)
#pragma unroll
        for (int jj = 0; jj < RK; ++jj)
#ifdef WG_ABL_NOMFMA
          asm volatile("" ::"v"(yf[i]), "v"(xf[jj]));
#else
          acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yf[i], xf[jj], acc[i][jj], 0, 0, 0);
#endif
    }
  }
  // D[n][k]: k = lane & 31, n = 8*(e>>2) + 4*(lane>>5) + (e&3)
  float* slab = a.part + (size_t)split * a.N * a.Kp;
#pragma unroll
  for (int i = 0; i < RN; ++i)
#pragma unroll
    for (int jj = 0; jj < RK; ++jj) {
      int k = k0 + (wk * RK + jj) * 32 + (lane & 31);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        int n = n0 + (wn * RN + i) * 32 + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3);
        if (n < n_hi && k < a.Kp) slab[(size_t)n * a.Kp + k] = acc[i][jj][e];
      }
    }
}

// ---- ROW3 variant: 3x3 / stride 1 / pad 1 (the CSP blocks' conv2, csp.py:30-46) -------------------------------------
// The generic kernel above gives every 128-column k tile its own block: dY is re-staged once per k tile (5x for a
// 64-channel layer) and every input pixel once per tap (9x) - these layers run 1.5-1.9x their forward time, bound by
// that L2 -> LDS fill, not by HBM or the MFMAs.  Here a block covers ALL NINE taps of a 32-channel chunk (x WC chunks):
// waves are laid out (n group) x (kernel row kh) x (chunk); per 32-row reduction step the dY tile is staged once for
// the whole block, and for each (kh, chunk) ONE run of 34 consecutive source pixels [m - 1, m + 32] of image row
// oy + kh - 1 - it serves the three taps kw = 0, 1, 2 as row offsets 0, 1, 2 of the transposed fragment reads.  Pixels
// are taken in flattened (b, y, x) order; a staged pixel whose image row is outside the image for this kh is an
// out-of-range buffer offset (zero-filled), and the left / right image border - where the flattened neighbour belongs
// to another image row - is a per-element mask on the X fragments of taps 0 and 2 (rows with ox == 0 / ox == W - 1,
// from a wave-uniform row bit mask; most steps have none).  Same slab layout as the generic kernel, so the same
// reduction kernels follow.
template <int WN, int RN, int WC, int NST>
__global__ __launch_bounds__(64 * WN * 3 * WC) void conv_wgrad_row3_kernel(WgradArgs a, uint32_t x_bytes, uint32_t dy_bytes) {
  constexpr int RS = 32;
  constexpr int NW = WN * 3 * WC;
  constexpr int TNB = WN * RN * 32;
  constexpr int RBY = TNB * 2;               // dY tile row bytes
  constexpr int RBX = 64;                    // X tile row bytes (32 channels): stride 64 B => conflict-free transposed reads
  constexpr int XR = 48;                     // staged X rows per (kh, chunk) tile (34 used): 3 DMA instructions x 16 rows
  constexpr int NIY = RS * RBY / 1024;
  constexpr int NXT = 3 * WC;                // X tiles per stage, tile = kh * WC + chunk
  constexpr int NIX = NXT * 3;
  constexpr int NI = NIY + NIX;
  constexpr int SLOTS = (NI + NW - 1) / NW;
  constexpr int XT_BYTES = XR * RBX;
  constexpr int STAGE = RS * RBY + NXT * XT_BYTES;
  constexpr int MY = swz_mask(RBY);
  constexpr int KS = RS / 16;
  static_assert((RS * RBY) % 1024 == 0 && XT_BYTES % 1024 == 0, "tiles must fill whole DMA instructions");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NST * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave % WC, kh = (wave / WC) % 3, wn = wave / (3 * WC);

  const int bid = blockIdx.x;
  const int tiles = a.tiles_n * a.tiles_k;
  int tile, split;
  if (!wg_map(a, bid, tiles, tile, split)) return;
  const int n0 = (tile % a.tiles_n) * TNB;
  const int c0 = (tile / a.tiles_n) * (WC * 32);
  const int m_begin = split * a.m_per_split;
  int m_end = m_begin + a.m_per_split;
  if (m_end > a.M) m_end = a.M;
  const int W = a.Ws, H = a.Hs, HW = a.Hs * a.Ws, BHW = a.B * HW;

#if defined(__HIP_DEVICE_COMPILE__)
  __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, dy_bytes, 0x00020000);
#endif

  // ---- this wave's DMA slots: instruction t = wave + i * NW; t < NIY feeds the dY tile, else X tile (t - NIY) / 3
  int s_row[SLOTS];                 // dY: tile row
  int s_p[SLOTS];                   // X: flattened source pixel of this lane's staged row at step 0 (may be negative)
  uint32_t s_off[SLOTS];            // dY: running byte offset; X: byte offset of (xcoff + channel)
  bool s_ok[SLOTS];
  int s_kh[SLOTS];
  int my_slots = 0;
#pragma unroll
  for (int i = 0; i < SLOTS; ++i) {
    const int t = wave + i * NW;
    s_row[i] = 0; s_p[i] = 0; s_off[i] = 0; s_ok[i] = false; s_kh[i] = 1;
    if (t >= NI) continue;
    ++my_slots;
    if (t < NIY) {
      const int L = t * 1024 + lane * 16;
      const int row = L / RBY, pc = (L % RBY) >> 4;
      const int c = pc ^ swz(row, MY);
      s_row[i] = row;
      s_ok[i] = (n0 + c * 8) < a.N;
      s_off[i] = (uint32_t)(((long)(m_begin + row) * a.ldy + a.ycoff + n0 + c * 8) * 2);
    } else {
      const int tx = t - NIY;
      const int xt = tx / 3, part = tx - xt * 3;
      const int tkh = xt / WC, twc = xt - tkh * WC;
      const int row = part * 16 + (lane >> 2);
      const int ch = c0 + twc * 32 + (lane & 3) * 8;
      s_kh[i] = tkh;
      s_ok[i] = ch < a.Cin;
      s_p[i] = m_begin - 1 + (tkh - 1) * W + row;
      s_off[i] = (uint32_t)((a.xcoff + ch) * 2);
    }
  }
  const uint32_t magic_hw = a.magic_hwo, magic_w = a.magic_wo;        // Ho = Hs, Wo = Ws for this geometry
  const uint32_t ldx2 = (uint32_t)a.ldx * 2u, ystep = (uint32_t)(RS * a.ldy * 2);

  int issued = 0;
  auto issue = [&](int stage) {
    unsigned char* Ys = lds + stage * STAGE;
    const int mb = m_begin + issued * RS;
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      const int t = wave + i * NW;
      if (t >= NI) continue;
      uint32_t vo;
      bool ok = s_ok[i];
      if (t < NIY) {
        ok = ok && (mb + s_row[i]) < m_end;
        vo = s_off[i];
        s_off[i] += ystep;
      } else {
        const int p = s_p[i];
        s_p[i] += RS;
        ok = ok && (unsigned)p < (unsigned)BHW;
        // image row of the staged pixel: exact magic divisions (estimate q or q + 1)
        uint32_t b = magic_hw ? __umulhi((uint32_t)p, magic_hw) : (uint32_t)p;
        int rem = p - (int)b * HW;
        if (rem < 0) rem += HW;
        uint32_t y = magic_w ? __umulhi((uint32_t)rem, magic_w) : (uint32_t)rem;
        if (rem - (int)y * W < 0) --y;
        // the output pixels this staged pixel serves lie in image row y + 1 - kh
        ok = ok && (s_kh[i] == 0 ? (int)y + 1 < H : (s_kh[i] == 2 ? y >= 1u : true));
        vo = (uint32_t)p * ldx2 + s_off[i];
      }
      vo = ok ? vo : 0xFFFFFFF0u;
#if defined(__HIP_DEVICE_COMPILE__)
      if (t < NIY)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, (__attribute__((address_space(3))) void*)(Ys + t * 1024), 16, vo, 0, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(Ys + RS * RBY + (t - NIY) * 1024), 16, vo, 0, 0, 0);
#else
      (void)vo; (void)Ys;
#endif
    }
    ++issued;
  };

  f32x16 acc[RN][3];
#pragma unroll
  for (int i = 0; i < RN; ++i)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][kw][e] = 0.f;

  const int tr_row = 8 * (lane >> 5) + ((lane & 15) >> 2);
  const int tr_col = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const int sy = swz(tr_row, MY);
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  uint32_t yoff[RN], xoff[3];
#pragma unroll
  for (int i = 0; i < RN; ++i) {
    const int col = (wn * RN + i) * 32 + tr_col;
    yoff[i] = (uint32_t)(tr_row * RBY + (((col >> 3) ^ sy) << 4) + (col & 7) * 2);
  }
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
    xoff[kw] = (uint32_t)(RS * RBY + (kh * WC + wc) * XT_BYTES + (tr_row + kw) * RBX + ((tr_col >> 3) << 4) + (tr_col & 7) * 2);

  const int nsteps = (m_end - m_begin + RS - 1) / RS;
  int ox_step = m_begin % W;                // ox of the step's first row (wave-uniform)
#pragma unroll
  for (int p = 0; p < NST - 1; ++p)
    if (p < nsteps) issue(p);
  for (int st = 0; st < nsteps; ++st) {
    const int newer = nsteps - 1 - st;
    if (newer >= NST - 2) {
      if (my_slots == SLOTS) wait_vm<(NST - 2) * SLOTS>(); else wait_vm<(NST - 2) * (SLOTS - 1)>();
    } else if (newer == 1 && NST > 3) {
      if (my_slots == SLOTS) wait_vm<SLOTS>(); else wait_vm<SLOTS - 1>();
    } else {
      wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();
    if (st + NST - 1 < nsteps) issue((st + NST - 1) % NST);
    // rows of this step on the left / right image border (taps kw = 0 / kw = 2 must not see their flattened neighbour)
    uint32_t inv0 = 0, inv2 = 0;
    for (int r = ox_step == 0 ? 0 : W - ox_step; r < RS; r += W) inv0 |= 1u << r;
    for (int r = W - 1 - ox_step; r < RS; r += W) inv2 |= 1u << r;
    ox_step += RS;
    while (ox_step >= W) ox_step -= W;

    const uint32_t sb = lds_base + (uint32_t)((st % NST) * STAGE);
    s16x4 ylo[KS][RN], yhi[KS][RN], xlo[KS][3], xhi[KS][3];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int i = 0; i < RN; ++i) {
        const uint32_t p = sb + yoff[i] + (uint32_t)(ks * 16 * RBY);
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(ylo[ks][i]) : "v"(p) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(yhi[ks][i]) : "v"(p + 4 * RBY) : "memory");
      }
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const uint32_t p = sb + xoff[kw] + (uint32_t)(ks * 16 * RBX);
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(xlo[ks][kw]) : "v"(p) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(xhi[ks][kw]) : "v"(p + 4 * RBX) : "memory");
      }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      constexpr int PER = 2 * (RN + 3);
      if (ks == KS - 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(PER < 15 ? PER : 15) : "memory");
#pragma unroll
      for (int i = 0; i < RN; ++i) asm volatile("" : "+v"(ylo[ks][i]), "+v"(yhi[ks][i]));
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) asm volatile("" : "+v"(xlo[ks][kw]), "+v"(xhi[ks][kw]));
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
      if ((inv0 | inv2) != 0) {      // wave-uniform: border rows in this step (1 step in W / 32 for the wide layers)
        const int sh = ks * 16 + 8 * (lane >> 5);
#pragma unroll
        for (int side = 0; side < 2; ++side) {
          const uint32_t bits = ((side == 0 ? inv0 : inv2) >> sh) & 0xFFu;     // element j of this lane's 8 rows
          const int kw = side == 0 ? 0 : 2;
          u32x2 lo = __builtin_bit_cast(u32x2, xlo[ks][kw]), hi = __builtin_bit_cast(u32x2, xhi[ks][kw]);
          lo[0] &= ((bits & 1u) ? 0u : 0xFFFFu) | ((bits & 2u) ? 0u : 0xFFFF0000u);
          lo[1] &= ((bits & 4u) ? 0u : 0xFFFFu) | ((bits & 8u) ? 0u : 0xFFFF0000u);
          hi[0] &= ((bits & 16u) ? 0u : 0xFFFFu) | ((bits & 32u) ? 0u : 0xFFFF0000u);
          hi[1] &= ((bits & 64u) ? 0u : 0xFFFFu) | ((bits & 128u) ? 0u : 0xFFFF0000u);
          xlo[ks][kw] = __builtin_bit_cast(s16x4, lo); xhi[ks][kw] = __builtin_bit_cast(s16x4, hi);
        }
      }
      bf16x8 yf[RN], xf[3];
#pragma unroll
      for (int i = 0; i < RN; ++i) {
        s16x8 t = {ylo[ks][i][0], ylo[ks][i][1], ylo[ks][i][2], ylo[ks][i][3], yhi[ks][i][0], yhi[ks][i][1], yhi[ks][i][2], yhi[ks][i][3]};
        yf[i] = __builtin_bit_cast(bf16x8, t);
      }
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        s16x8 t = {xlo[ks][kw][0], xlo[ks][kw][1], xlo[ks][kw][2], xlo[ks][kw][3], xhi[ks][kw][0], xhi[ks][kw][1], xhi[ks][kw][2], xhi[ks][kw][3]};
        xf[kw] = __builtin_bit_cast(bf16x8, t);
      }
#pragma unroll
      for (int i = 0; i < RN; ++i)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
          acc[i][kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yf[i], xf[kw], acc[i][kw], 0, 0, 0);
    }
  }
  // D[n][k]: k = lane & 31, n = 8*(e>>2) + 4*(lane>>5) + (e&3); slab column k = (kh*3 + kw) * Cin + channel
  float* slab = a.part + (size_t)split * a.N * a.Kp;
  const int ch = c0 + wc * 32 + (lane & 31);
#pragma unroll
  for (int i = 0; i < RN; ++i)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int k = (kh * 3 + kw) * a.Cin + ch;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = n0 + (wn * RN + i) * 32 + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3);
        if (n < a.N && ch < a.Cin) slab[(size_t)n * a.Kp + k] = acc[i][kw][e];
      }
    }
}

// grad[n][ci][kh][kw] = scale * sum_s part[s][n][k(kh,kw,ci)]   (stem: k = (kh, kw', dx, c4), see pack)
// block = RK consecutive k x RL split lanes (RK * RL = 256): each lane streams RK*4-byte contiguous pieces of its
// slabs with four independent partial sums (loads in flight), fixed-order LDS combine => deterministic.
constexpr int RED_K = 32, RED_L = 8;    // measured: 16x16 0.65 ms, 32x8 0.45 ms, 64x4 0.50 ms, 128x2 0.71 ms per step
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* part, float* grad, int splits, int Nfull,
                                                           int N, int K, int Kp, int Cin, int KK, int stem,
                                                           float scale, float* grad2 = nullptr, int n_first = 1 << 30) {
  __shared__ float sm[RED_L][RED_K + 1];
  const int kx = threadIdx.x % RED_K, sl = threadIdx.x / RED_K;
  const long idx = (long)blockIdx.x * RED_K + kx;
  const long total = (long)N * K;
  float s = 0.f;
  int n = 0, k = 0;
  if (idx < total) {
    n = (int)(idx / K);
    k = (int)(idx - (long)n * K);
    const float* p = part + (size_t)n * Kp + k;
    const size_t slab = (size_t)Nfull * Kp;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int i = sl;
    for (; i + 3 * RED_L < splits; i += 4 * RED_L) {
      s0 += p[(size_t)i * slab];
      s1 += p[(size_t)(i + RED_L) * slab];
      s2 += p[(size_t)(i + 2 * RED_L) * slab];
      s3 += p[(size_t)(i + 3 * RED_L) * slab];
    }
    for (; i < splits; i += RED_L) s0 += p[(size_t)i * slab];
    s = (s0 + s1) + (s2 + s3);
  }
  sm[sl][kx] = s;
  __syncthreads();
  if (sl == 0 && idx < total) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < RED_L; ++i) t += sm[i][kx];
    if (stem) {
      // k = (kh*3 + kwp)*8 + dx*4 + c, real weight [n][c(3)][kh(6)][kw = 2*kwp+dx (6)]
      int c = k & 3, dx = (k >> 2) & 1, tt = k >> 3;
      int kh = tt / 3, kwp = tt - kh * 3;
      if (c < 3) grad[(size_t)n * 108 + c * 36 + kh * 6 + 2 * kwp + dx] = t * scale;
    } else {
      int tap = k / Cin;
      int ci = k - tap * Cin;
      float* dst = n >= n_first ? grad2 : grad;          // dual form: slab rows n_first .. belong to the second layer
      if (n >= n_first) n -= n_first;
      dst[(size_t)n * Cin * KK + ci * KK + tap] = t * scale;
    }
  }
}

// The same reduction with 16-byte accesses: a thread owns FOUR consecutive k of one slab row (K and Kp are multiples of 4, so
// a quad never crosses rows) for its split lane; per (n, k) the additions and their order are exactly wgrad_reduce_kernel's
// (four chains over the lane's splits, (s0 + s1) + (s2 + s3), then the eight lanes in order) => bit-identical gradients,
// a quarter of the load instructions and 4 x the bytes in flight per thread.  KODHIP_WGRAD_REDUCE_V4=0: the scalar kernel.
constexpr int RED_Q = 32;                // k quads per block (128 consecutive k)
__global__ __launch_bounds__(256) void wgrad_reduce_v4_kernel(const float* part, float* grad, int splits, int Nfull,
                                                              int N, int K, int Kp, int Cin, int KK, int stem,
                                                              float scale, float* grad2 = nullptr, int n_first = 1 << 30) {
  __shared__ f32x4 sm[RED_L][RED_Q];
  const int qx = threadIdx.x % RED_Q, sl = threadIdx.x / RED_Q;
  const int KQ = K >> 2;
  const long q = (long)blockIdx.x * RED_Q + qx;
  const long total = (long)N * KQ;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  int n = 0, k = 0;
  if (q < total) {
    n = (int)(q / KQ);
    k = (int)(q - (long)n * KQ) * 4;
    const float* p = part + (size_t)n * Kp + k;
    const size_t slab = (size_t)Nfull * Kp;
    f32x4 s0 = s, s1 = s, s2 = s, s3 = s;
    int i = sl;
    for (; i + 3 * RED_L < splits; i += 4 * RED_L) {
      const f32x4 a0 = kod_load_once<f32x4>(p + (size_t)i * slab);
      const f32x4 a1 = kod_load_once<f32x4>(p + (size_t)(i + RED_L) * slab);
      const f32x4 a2 = kod_load_once<f32x4>(p + (size_t)(i + 2 * RED_L) * slab);
      const f32x4 a3 = kod_load_once<f32x4>(p + (size_t)(i + 3 * RED_L) * slab);
      s0 += a0; s1 += a1; s2 += a2; s3 += a3;
    }
    for (; i < splits; i += RED_L) s0 += kod_load_once<f32x4>(p + (size_t)i * slab);
    s = (s0 + s1) + (s2 + s3);
  }
  sm[sl][qx] = s;
  __syncthreads();
  if (sl == 0 && q < total) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < RED_L; ++i) t += sm[i][qx];
    float* dst = n >= n_first ? grad2 : grad;            // dual form: slab rows n_first .. belong to the second layer
    if (n >= n_first) n -= n_first;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int kk = k + e;
      if (stem) {
        const int c = kk & 3, dx = (kk >> 2) & 1, tt = kk >> 3;
        const int kh = tt / 3, kwp = tt - kh * 3;
        if (c < 3) dst[(size_t)n * 108 + c * 36 + kh * 6 + 2 * kwp + dx] = t[e] * scale;
      } else {
        const int tap = kk / Cin;
        const int ci = kk - tap * Cin;
        dst[(size_t)n * Cin * KK + ci * KK + tap] = t[e] * scale;
      }
    }
  }
}

static int launch_wgrad_reduce(const float* part, float* grad, int splits, int Nfull, int N, int K, int Kp, int Cin, int KK,
                               int stem, float scale, float* grad2, int n_first, hipStream_t stream) {
  static int v4 = -1;
  if (v4 < 0) { const char* e = getenv("KODHIP_WGRAD_REDUCE_V4"); v4 = e ? atoi(e) : 1; }
  if (v4 && K % 4 == 0 && Kp % 4 == 0 && (reinterpret_cast<uintptr_t>(part) & 15) == 0)
    hipLaunchKernelGGL(wgrad_reduce_v4_kernel, dim3(cdiv((long)N * (K >> 2), RED_Q)), dim3(256), 0, stream, part, grad, splits, Nfull, N, K,
                       Kp, Cin, KK, stem, scale, grad2, n_first);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv((long)N * K, RED_K)), dim3(256), 0, stream, part, grad, splits, Nfull, N, K, Kp, Cin,
                       KK, stem, scale, grad2, n_first);
  KOD_LAUNCH_CHECK("wgrad_reduce");
  return KOD_OK;
}

// ---- the same reduction for MANY layers in one launch.  A training step has ~60 weight gradients; their slab
// reductions are 5-30 us kernels of a few hundred blocks each (1.0 ms per step as separate launches, mostly launch
// ramps and tails).  Every layer keeps its own slab region, and ONE launch per gradient bucket reduces them all: block
// -> layer by binary search over a small descriptor table, then exactly wgrad_reduce_kernel's arithmetic (same
// fixed-order sums => bit-identical gradients).
struct ReduceDesc {
  long part_off, grad_off;      // in floats, relative to the partials / gradient arenas
  int splits, Nfull, N, K, Kp, Cin, KK, stem;
  float scale;
  int block_start;              // first block of this layer; blocks = cdiv(N * K, RED_K)
};

__global__ __launch_bounds__(256) void wgrad_reduce_batched_kernel(const float* parts, float* grads, const ReduceDesc* descs,
                                                                   int n_desc) {
  __shared__ float sm[RED_L][RED_K + 1];
  int lo = 0, hi = n_desc - 1;
  const int bid = blockIdx.x;
  while (lo < hi) {             // last descriptor whose block_start <= bid (block-uniform: scalar loads)
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].block_start <= bid) lo = mid; else hi = mid - 1;
  }
  const ReduceDesc d = descs[lo];
  const float* part = parts + d.part_off;
  float* grad = grads + d.grad_off;
  const int kx = threadIdx.x % RED_K, sl = threadIdx.x / RED_K;
  const long idx = (long)(bid - d.block_start) * RED_K + kx;
  const long total = (long)d.N * d.K;
  float s = 0.f;
  int n = 0, k = 0;
  if (idx < total) {
    n = (int)(idx / d.K);
    k = (int)(idx - (long)n * d.K);
    const float* p = part + (size_t)n * d.Kp + k;
    const size_t slab = (size_t)d.Nfull * d.Kp;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int i = sl;
    for (; i + 3 * RED_L < d.splits; i += 4 * RED_L) {
      s0 += p[(size_t)i * slab];
      s1 += p[(size_t)(i + RED_L) * slab];
      s2 += p[(size_t)(i + 2 * RED_L) * slab];
      s3 += p[(size_t)(i + 3 * RED_L) * slab];
    }
    for (; i < d.splits; i += RED_L) s0 += p[(size_t)i * slab];
    s = (s0 + s1) + (s2 + s3);
  }
  sm[sl][kx] = s;
  __syncthreads();
  if (sl == 0 && idx < total) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < RED_L; ++i) t += sm[i][kx];
    if (d.stem) {
      int c = k & 3, dx = (k >> 2) & 1, tt = k >> 3;
      int kh = tt / 3, kwp = tt - kh * 3;
      if (c < 3) grad[(size_t)n * 108 + c * 36 + kh * 6 + 2 * kwp + dx] = t * d.scale;
    } else {
      int tap = k / d.Cin;
      int ci = k - tap * d.Cin;
      grad[(size_t)n * d.Cin * d.KK + ci * d.KK + tap] = t * d.scale;
    }
  }
}

#ifndef WG_NST
#define WG_NST 4
#endif
#ifndef WG_RS
#define WG_RS 32
#endif
constexpr int WGRAD_STAGES = WG_NST;     // LDS ring depth
constexpr int WGRAD_ROWS = WG_RS;        // reduction rows per ring stage

template <int WN, int WK, int RN, int RK>
int launch_cfg(WgradArgs a, hipStream_t stream) {
  constexpr int TNB = WN * RN * 32, TKB = WK * RK * 32;
  a.tiles_n = a.n_half ? 2 * cdiv(a.n_half, TNB) : cdiv(a.N, TNB);
  a.tiles_k = cdiv(a.Kp, TKB);
  int grid = wg_grid(a);
  // LDS-DMA ring by default: +4-5 % on the whole training step over the register-staged kernel (in the network the
  // operands come from HBM and the deeper prefetch pays; back-to-back microbenchmarks with L2-hot operands show
  // mixed results for the 3x3 layers because an LDS-DMA piece costs 60+ issue cycles).  KODHIP_WGRAD_DMA=none
  // selects the register-staged kernel for A/B runs; an operand that does not fit a 32-bit buffer range always
  // takes it.
  const long xb = (long)a.B * a.Hs * a.Ws * a.ldx * 2, yb = (long)a.M * a.ldy * 2;
  static const char* mode = getenv("KODHIP_WGRAD_DMA");
  if (!(mode && mode[0] == 'n') && xb < (1l << 32) - 64 && yb < (1l << 32) - 64) {
    const bool pw = a.KH == 1 && a.KW == 1 && a.SH == 1 && a.SW == 1 && a.PH == 0 && a.PW == 0;
    const dim3 g(grid), b(64 * WN * WK);
    if (pw) hipLaunchKernelGGL((conv_wgrad_dma_kernel<WN, WK, RN, RK, WGRAD_STAGES, WGRAD_ROWS, 1>), g, b, 0, stream, a, (uint32_t)xb, (uint32_t)yb);
    else hipLaunchKernelGGL((conv_wgrad_dma_kernel<WN, WK, RN, RK, WGRAD_STAGES, WGRAD_ROWS, 0>), g, b, 0, stream, a, (uint32_t)xb, (uint32_t)yb);
    KOD_LAUNCH_CHECK("conv_wgrad_dma");
    return KOD_OK;
  }
  hipLaunchKernelGGL((conv_wgrad_kernel<WN, WK, RN, RK>), dim3(grid), dim3(64 * WN * WK), 0, stream, a);
  KOD_LAUNCH_CHECK("conv_wgrad");
  return KOD_OK;
}

// ---- the stem's backward in one kernel: dY = f(dA, y) formed on the fly, dW slabs out ---------------------------------
// The stem (6x6 / stride 2 / pad 2 on the image, here a 6x3 / stride (2,1) conv over 8-value pixel PAIRS) has no data
// gradient, so its dY = k1 * dA * silu'(z) + k2 * y + k3 has exactly one reader: this weight gradient.  The separate
// BatchNorm/SiLU backward pass read dA and y (2 x 0.42 GB at B = 64 / 640 px), wrote dY (0.42 GB) and the generic weight
// gradient read it back and gathered every pixel pair 18 x from L2 (fill-bound: 233 us for 0.63 GB).  Here a block walks
// row-aligned tiles of TW output pixels.  Everything a tile needs goes HBM -> LDS by DMA, one tile ahead of the
// arithmetic: the six pair-row runs (TW + 2 pairs each, the two border pairs zero-filled by out-of-range offsets - no tap
// masks; they serve all 18 taps through per-lane fragment addresses, run row i + kw') and the dA / y tiles.  dY is formed
// IN LDS with the arithmetic of bn_silu_bwd_apply_kernel (same bf16 rounding), over the dA tile, which then is the
// [m][n] operand; wave w owns k columns 32 w .. 32 w + 31 of the 160-column slab.  dY never exists in HBM.
// Every LDS access is inline asm: with compiler-visible LDS writes or register loads next to the DMA the compiler's own
// waits drain the prefetch in front of them (measured: movement 256 us + transform 98 us + MFMA 80 us ran back to back).
struct StemBwdArgs {
  const bf16_t* x;            // pixel pairs [B][Hs][Wp][8]
  const bf16_t* dA; int lda, dacoff;
  const bf16_t* y; int ldy;
  const float* scale; const float* shift; const float* coef;      // coef: k1[N] | k2[N] | k3[N]
  float* part;                // [blocks][32][160]
  int B, Hs, Wp, Ho, Wo, N;
  int tiles_per_row, tiles, tiles_per_block;
  uint32_t x_bytes, da_bytes, y_bytes;
};

// staged pairs per kernel row: TW + 2 used, and PRW * 16 B = 128 mod 256 so that the two kernel rows a 4-tap column group
// touches land on disjoint LDS banks
__host__ __device__ constexpr int stem_prw(int TW) { return TW == 160 ? 168 : 88; }
__host__ __device__ constexpr int stem_xb(int TW) { return (6 * stem_prw(TW) * 16 + 1023) / 1024 * 1024; }

// NT = 32-channel output tiles per block: 1 for N <= 32 (yv5n / yv5s), 2 for N <= 64 (yv5m: 48, yv5l: 64) - the dA / y
// tile rows are then 128 B (two-way bank conflicts in the transposing reads of dY, accepted) and every wave runs two MFMAs
// per staged X fragment.
template <int TW, int NT>
__global__ __launch_bounds__(320, NT == 2 ? 3 : (TW == 160 ? 3 : 5)) void conv_stem_bwd_fused_kernel(StemBwdArgs a) {
  constexpr int PRW = stem_prw(TW), NW = 5;
  constexpr int CH = 4 * NT, RBG = 64 * NT;              // 16-byte chunks / bytes per dA (dY) / y tile row
  constexpr int XB = stem_xb(TW), GB = TW * RBG;         // X runs | dA tile (becomes dY) | y tile
  constexpr int STAGE = XB + 2 * GB;
  constexpr int NIX = XB / 1024, NIG = GB / 1024, NI = NIX + 2 * NIG;     // DMA instructions per tile (1 KB each)
  constexpr int ITEMS = TW * CH / 320;                   // 16-byte (pixel, 8-channel chunk) items per thread
  static_assert(GB % 1024 == 0 && TW * CH % 320 == 0 && 320 % CH == 0, "tile must fill whole DMA instructions / thread items");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t_begin = blockIdx.x * a.tiles_per_block;
  int t_end = t_begin + a.tiles_per_block;
  if (t_end > a.tiles) t_end = a.tiles;

#if defined(__HIP_DEVICE_COMPILE__)
  __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)a.dA, 0, a.da_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, a.y_bytes, 0x00020000);
#endif

  // per-thread constants of the dY arithmetic: this thread's items are (pixel q / CH, channel chunk q % CH), q = tid + 320 j
  // (320 is a multiple of CH: the chunk is the same for all of a thread's items)
  const int cc = tid % CH;
  float sc[8], sh[8], k1[8], k2[8], k3[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = cc * 8 + e;
    const bool ok = c < a.N;
    sc[e] = ok ? a.scale[c] : 0.f; sh[e] = ok ? a.shift[c] : 0.f;
    k1[e] = ok ? a.coef[c] : 0.f; k2[e] = ok ? a.coef[a.N + c] : 0.f; k3[e] = ok ? a.coef[2 * a.N + c] : 0.f;
  }
  const bool ch_ok = cc * 8 < a.N;

  // one tile's DMA: instruction i = wave + 5 k; i < NIX: pair-row runs (LDS slot 64 i + lane = kh * PRW + r), then the dA
  // tile, then the y tile (slot = CH * pixel + chunk)
  auto stage = [&](int t, int buf) {
    const int row = t / a.tiles_per_row;                  // b * Ho + oy
    const int ox0 = (t - row * a.tiles_per_row) * TW;
    const int b = row / a.Ho, oy = row - b * a.Ho;
    unsigned char* S = lds + buf * STAGE;
#pragma unroll
    for (int k = 0; k < (NI + NW - 1) / NW; ++k) {
      const int i = wave + NW * k;
      if (i >= NI) continue;
      uint32_t vo = 0xFFFFFFF0u;
      if (i < NIX) {
        const int L = i * 64 + lane;
        const int kh = L / PRW, r = L - kh * PRW;
        const int iy = 2 * oy - 2 + kh, px = ox0 - 1 + r;
        if (kh < 6 && r < TW + 2 && (unsigned)iy < (unsigned)a.Hs && (unsigned)px < (unsigned)a.Wp)
          vo = (uint32_t)((((long)b * a.Hs + iy) * a.Wp + px) * 16);
      } else {
        const bool is_g = i < NIX + NIG;
        const int q = (i - NIX - (is_g ? 0 : NIG)) * 64 + lane;
        const int p = q / CH, c = q % CH;
        const long m = (long)row * a.Wo + ox0 + p;
        if (ox0 + p < a.Wo && c * 8 < a.N)
          vo = is_g ? (uint32_t)((m * a.lda + a.dacoff + c * 8) * 2) : (uint32_t)((m * a.ldy + c * 8) * 2);
#ifdef KOD_ABL_STEMY      // (tools/build_ablate.sh: what the pre-BN tensor's read costs this kernel - y from one hot 4 KB instead)
        if (!is_g) vo &= 0xFFFu;
#endif
      }
#if defined(__HIP_DEVICE_COMPILE__)
      if (i < NIX)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(S + i * 1024), 16, vo, 0, 0, 0);
      else if (i < NIX + NIG)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (__attribute__((address_space(3))) void*)(S + i * 1024), 16, vo, 0, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, (__attribute__((address_space(3))) void*)(S + i * 1024), 16, vo, 0, 0, 0);
#else
      (void)vo; (void)S;
#endif
    }
  };

  f32x16 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;

  // fragment addresses (transposing reads, lane geometry of conv_wgrad_dma_kernel): rows = reduction index m
  const int tr_row = 8 * (lane >> 5) + ((lane & 15) >> 2);
  const int tr_col = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  const uint32_t yoff = (uint32_t)(XB + tr_row * RBG + tr_col * 2);
  int tap = wave * 4 + (tr_col >> 3);
  if (tap > 17) tap = 17;                                   // slab columns 144 .. 159 are padding (never reduced)
  const int tkh = tap / 3, tkw = tap - tkh * 3;
  const uint32_t xoff = (uint32_t)((tkh * PRW + tkw + tr_row) * 16 + (tr_col & 7) * 2);

  if (t_begin < t_end) stage(t_begin, 0);
  int buf = 0;
  for (int t = t_begin; t < t_end; ++t) {
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();       // this tile has landed (every wave's share); the other stage's readers are done
    if (t + 1 < t_end) stage(t + 1, buf ^ 1);

    // ---- dY of this tile, in place over the dA tile
    const uint32_t sb = lds_base + (uint32_t)(buf * STAGE);
    const int ox0 = (t % a.tiles_per_row) * TW;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      const int q = tid + 320 * j, p = q / CH;
      const uint32_t ga = sb + XB + (uint32_t)q * 16, ya = ga + GB;
      u32x4 gr, yr;
      asm volatile("ds_read_b128 %0, %1" : "=v"(gr) : "v"(ga) : "memory");
      asm volatile("ds_read_b128 %0, %1" : "=v"(yr) : "v"(ya) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      asm volatile("" : "+v"(gr), "+v"(yr));
      const bf16x8 g8 = __builtin_bit_cast(bf16x8, gr);
      const bf16x8 y8 = __builtin_bit_cast(bf16x8, yr);
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float yv = (float)y8[e];                       // = bn_silu_bwd_apply_kernel's arithmetic
        const float z = __builtin_fmaf(yv, sc[e], sh[e]);
        const float sg = kod_sigmoid_l2(KOD_NEG_LOG2E * z);
        const float dz = kod_silu_bwd((float)g8[e], z, sg);
        o[e] = (bf16_t)__builtin_fmaf(k1[e], dz, __builtin_fmaf(k2[e], yv, k3[e]));
      }
      if (!(ch_ok && ox0 + p < a.Wo)) o = bf16x8{};        // ragged last tile of a row / channels past N: zero rows
      const u32x4 ov = __builtin_bit_cast(u32x4, o);
      asm volatile("ds_write_b128 %0, %1" ::"v"(ga), "v"(ov) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

#pragma unroll 2
    for (int ks = 0; ks < TW / 16; ++ks) {
      s16x4 ylo[NT], yhi[NT], xlo, xhi;
      const uint32_t px = sb + xoff + (uint32_t)(ks * 16 * 16);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const uint32_t py = sb + yoff + (uint32_t)(ks * 16 * RBG + n * 64);
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(ylo[n]) : "v"(py) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(yhi[n]) : "v"(py + 4 * RBG) : "memory");
      }
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(xlo) : "v"(px) : "memory");
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(xhi) : "v"(px + 4 * 16) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      asm volatile("" : "+v"(xlo), "+v"(xhi));
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      const s16x8 tx = {xlo[0], xlo[1], xlo[2], xlo[3], xhi[0], xhi[1], xhi[2], xhi[3]};
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        asm volatile("" : "+v"(ylo[n]), "+v"(yhi[n]));
        const s16x8 ty = {ylo[n][0], ylo[n][1], ylo[n][2], ylo[n][3], yhi[n][0], yhi[n][1], yhi[n][2], yhi[n][3]};
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ty), __builtin_bit_cast(bf16x8, tx), acc[n], 0, 0, 0);
      }
    }
    buf ^= 1;
  }

  // D[n][k]: k = 32 wave + (lane & 31), n = 32 tile + 8 (e >> 2) + 4 (lane >> 5) + (e & 3)
  float* slab = a.part + (size_t)blockIdx.x * (32 * NT) * 160;
  const int k = wave * 32 + (lane & 31);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int n = nt * 32 + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3);
      slab[(size_t)n * 160 + k] = acc[nt][e];
    }
}

// ROW3 form (conv_wgrad_row3_kernel): 3x3 / stride 1 / pad 1 with whole 32-channel chunks.
// KODHIP_WGRAD_ROW3: 0 = off, 1 (default) = where it measured faster, 2 = every eligible layer.
// Measured at B = 64 / 640 px (profiles/r03_convbench.txt, generic -> ROW3): 32->32 @160 138 -> 125 us, 256->256 @20
// 93 -> 77 us, but 64->64 @80 92 -> 104 us and 128->128 @40 58 -> 75 us: a block that owns all nine taps has a 4.5x larger
// output tile, so the same number of resident blocks writes 4.5x the split-K slab bytes (75 MB against 15-33 MB for
// the two middle layers, whose whole operand traffic is 52-105 MB).  It pays where the slab is small against the
// operands (N <= 32) or where the generic kernel's tile count leaves it few rows per block (Cin >= 256).
struct Row3Cfg { bool on; int wn, rn, wc, tnb; };
Row3Cfg row3_cfg(int N, int Cin, int KH, int KW, int SH, int SW, int PH, int PW) {
  static int mode = -1;
  if (mode < 0) { const char* e = getenv("KODHIP_WGRAD_ROW3"); mode = e ? atoi(e) : 1; }
  Row3Cfg c = {false, 0, 0, 0, 0};
  if (!mode || KH != 3 || KW != 3 || SH != 1 || SW != 1 || PH != 1 || PW != 1 || Cin % 32 != 0 || N % 8 != 0) return c;
  // mode 1 (default): where the form measured faster than the generic split-K kernel IN THE STEP: narrow outputs (N <= 32) and
  // Cin >= 256 as long as the nine-tap output tile stays small (N <= 256: 256 -> 256 @20 79 vs 94 us; yv5m's 384 -> 384 @20
  // 256 vs 198 us - round 3's rule took the form there).  (96 -> 96 @80 gains alone, 174 vs 197 us, and loses in the yv5m step:
  // 2 220 vs 2 232 img/s - its larger slabs cost the co-running main chain more than the kernel saves.)
  if (mode == 1 && !(N <= 32 || (Cin >= 256 && N <= 256))) return c;
  c.on = true;
  if (N <= 32) { c.wn = 1; c.rn = 1; }
  else if (N <= 64) { c.wn = 1; c.rn = 2; }
  else { c.wn = 2; c.rn = 2; }
  c.wc = Cin >= 64 ? 2 : 1;
  c.tnb = c.wn * c.rn * 32;
  return c;
}

template <int WN, int RN, int WC>
int launch_row3(WgradArgs a, hipStream_t stream) {
  // ring depth: 4 stages where two 6-wave / one 12-wave block still fit a CU's 160 KB, 3 stages for the 3-wave blocks
  // (4 resident) and the widest 6-wave stage
  constexpr int NST = (WN * 3 * WC == 3 || (WN == 1 && RN == 2 && WC == 2)) ? 3 : 4;
  const long xb = (long)a.B * a.Hs * a.Ws * a.ldx * 2, yb = (long)a.M * a.ldy * 2;
  const int grid = wg_grid(a);
  hipLaunchKernelGGL((conv_wgrad_row3_kernel<WN, RN, WC, NST>), dim3(grid), dim3(64 * WN * 3 * WC), 0, stream, a, (uint32_t)xb, (uint32_t)yb);
  KOD_LAUNCH_CHECK("conv_wgrad_row3");
  return KOD_OK;
}

void tile_shape(int N, int Kp, int* tn, int* tk) {
  *tn = N > 64 ? 128 : (N > 32 ? 64 : 32);
  // yv5m widths: N = 192 as three 64-row tiles (all used) instead of two 128-row tiles (a quarter idle); KODHIP_WGRAD_TN192: A/B knob
  // (measured, profiles/r04_convbench_yv5m_tiles.txt: 192 -> 192 3x3 @40 234 -> 191 us, 96 -> 192 s2 431 -> 337, 384 -> 192 1x1 49 -> 34)
  // (N = 96 as three 32-row tiles instead of one 128-row tile: 198 -> 296 us, not taken)
  static int tn192 = -1;
  if (tn192 < 0) { const char* e = getenv("KODHIP_WGRAD_TN192"); tn192 = e ? atoi(e) : 64; }
  if (N > 128 && N <= 192 && tn192 == 64) *tn = 64;
  *tk = Kp > 64 ? 128 : (Kp > 32 ? 64 : 32);
  // narrow outputs: let one block cover all of K so dY is streamed once
  if (*tn == 32 && (Kp == 160 || Kp == 288)) *tk = Kp;
}

}  // namespace

extern "C" {

static int wgrad_splits_target(long M, int N, int Kp) {
  int tn, tk;
  tile_shape(N, Kp, &tn, &tk);
  int tiles = cdiv(N, tn) * cdiv(Kp, tk);
  static int slots = 0;                  // resident-block target (KODHIP_WGRAD_SLOTS: A/B knob)
  if (!slots) { const char* e = getenv("KODHIP_WGRAD_SLOTS"); slots = e ? atoi(e) : 512; if (slots < 8) slots = 512; }
  int s = slots / tiles;
  if (s < 1) s = 1;
  long maxs = (M + 255) / 256;
  if (s > maxs) s = (int)maxs;
  if (s < 1) s = 1;
  return s;
}

static int wgrad_rows_per_split(long M, int N, int Kp) {
  return cdiv(cdiv(M, wgrad_splits_target(M, N, Kp)), 32) * 32;
}

// rows per split for a given geometry: the ROW3 form has its own block count (tiles = n tiles x channel chunks)
static int wgrad_rows_per_split_geo(long M, int N, int Cin, int KH, int KW, int SH, int SW, int PH, int PW, int Kp,
                                    long x_bytes, long dy_bytes) {
  const Row3Cfg c = row3_cfg(N, Cin, KH, KW, SH, SW, PH, PW);
  if (!c.on || x_bytes >= (1l << 32) - 64 || dy_bytes >= (1l << 32) - 64) return wgrad_rows_per_split(M, N, Kp);
  const int tiles = cdiv(N, c.tnb) * cdiv(Cin, c.wc * 32);
  static int wave_slots = 0;                            // KODHIP_WGRAD_ROW3_SLOTS: resident-wave target (A/B knob; default 3072)
  if (!wave_slots) { const char* e = getenv("KODHIP_WGRAD_ROW3_SLOTS"); wave_slots = e ? atoi(e) : 3072; if (wave_slots < 96) wave_slots = 3072; }
  const int slots = wave_slots / (c.wn * 3 * c.wc);    // resident blocks: 1024 (3 waves), 512 (6), 256 (12)
  int sp = slots / tiles;
  if (sp < 1) sp = 1;
  const long maxs = (M + 255) / 256;
  if (sp > maxs) sp = (int)maxs;
  if (sp < 1) sp = 1;
  return cdiv(cdiv(M, sp), 32) * 32;
}

// Exact split count for a layer geometry (ldx / ldy: row strides of the operands, for the 32-bit range test).
int kodhip_conv_wgrad_splits_geo(int B, int H, int W, int ldx, int Cin, int N, int KH, int KW, int SH, int SW, int PH, int PW,
                                 int Kp, int ldy) {
  const int Ho = (H + 2 * PH - KH) / SH + 1, Wo = (W + 2 * PW - KW) / SW + 1;
  const long M = (long)B * Ho * Wo;
  const long xb = (long)B * H * W * ldx * 2, yb = M * ldy * 2;
  return (int)cdiv(M, (long)wgrad_rows_per_split_geo(M, N, Cin, KH, KW, SH, SW, PH, PW, Kp, xb, yb));
}

// Number of reduction splits the launcher uses (exact; the partials region must hold splits * N * Kp floats).
int kodhip_conv_wgrad_splits(long M, int N, int Kp) { return (int)cdiv(M, (long)wgrad_rows_per_split(M, N, Kp)); }

static int wgrad_partial(WgradArgs& a, const void* x, const void* dy, float* partials,
                         int B, int H, int W, int ldx, int xcoff, int Cin,
                         int N, int KH, int KW, int SH, int SW, int PH, int PW, int Kp,
                         int ldy, int ycoff, hipStream_t stream) {
  KOD_CHECK_ARG(x && dy && partials, "conv_wgrad: null pointer");
  KOD_CHECK_ARG(Cin % 8 == 0 && ldx % 8 == 0 && xcoff % 8 == 0 && xcoff + Cin <= ldx, "conv_wgrad: bad input slice");
  KOD_CHECK_ARG(N % 8 == 0 && ldy % 8 == 0 && ycoff % 8 == 0 && ycoff + N <= ldy, "conv_wgrad: bad dy slice (N=%d ldy=%d)", N, ldy);
  KOD_CHECK_ARG(Kp % 32 == 0 && Kp >= KH * KW * Cin, "conv_wgrad: bad Kp");
  a = WgradArgs{};
  a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dy; a.part = partials;
  a.B = B; a.Hs = H; a.Ws = W; a.ldx = ldx; a.xcoff = xcoff; a.Cin = Cin;
  a.Ho = (H + 2 * PH - KH) / SH + 1; a.Wo = (W + 2 * PW - KW) / SW + 1;
  long M = (long)B * a.Ho * a.Wo;
  KOD_CHECK_ARG(M < (1l << 31) && (long)B * H * W < (1l << 31), "conv_wgrad: pixel count overflows int32");
  a.M = (int)M; a.N = N; a.K = KH * KW * Cin; a.Kp = Kp;
  a.KH = KH; a.KW = KW; a.SH = SH; a.SW = SW; a.PH = PH; a.PW = PW; a.ldy = ldy; a.ycoff = ycoff;
  a.magic_cin = magic_u32((uint32_t)Cin); a.magic_kw = magic_u32((uint32_t)KW);
  a.magic_hwo = magic_u32((uint32_t)(a.Ho * a.Wo)); a.magic_wo = magic_u32((uint32_t)a.Wo);
  const long xb = (long)B * H * W * ldx * 2, yb = M * ldy * 2;
  a.m_per_split = wgrad_rows_per_split_geo(M, N, Cin, KH, KW, SH, SW, PH, PW, Kp, xb, yb);
  a.splits = cdiv(M, a.m_per_split);
  const Row3Cfg r3 = row3_cfg(N, Cin, KH, KW, SH, SW, PH, PW);
  if (r3.on && xb < (1l << 32) - 64 && yb < (1l << 32) - 64) {
    a.tiles_n = cdiv(N, r3.tnb);
    a.tiles_k = cdiv(Cin, r3.wc * 32);
    if (r3.wn == 1 && r3.rn == 1 && r3.wc == 1) return launch_row3<1, 1, 1>(a, stream);
    if (r3.wn == 1 && r3.rn == 1) return launch_row3<1, 1, 2>(a, stream);
    if (r3.wn == 1 && r3.wc == 1) return launch_row3<1, 2, 1>(a, stream);
    if (r3.wn == 1) return launch_row3<1, 2, 2>(a, stream);
    if (r3.wc == 1) return launch_row3<2, 2, 1>(a, stream);
    return launch_row3<2, 2, 2>(a, stream);
  }
  int tn, tk, rc;
  tile_shape(N, Kp, &tn, &tk);
  if (tn == 128 && tk == 128) rc = launch_cfg<2, 2, 2, 2>(a, stream);
  else if (tn == 64 && tk == 128) rc = launch_cfg<2, 2, 1, 2>(a, stream);
  else if (tn == 32 && tk == 160) rc = launch_cfg<1, 5, 1, 1>(a, stream);
  else if (tn == 32 && tk == 288) rc = launch_cfg<1, 9, 1, 1>(a, stream);
  else if (tn == 32 && tk == 128) rc = launch_cfg<1, 4, 1, 1>(a, stream);
  else if (tn == 128 && tk == 64) rc = launch_cfg<2, 2, 2, 1>(a, stream);
  else if (tn == 64 && tk == 64) rc = launch_cfg<2, 2, 1, 1>(a, stream);
  else if (tn == 32 && tk == 64) rc = launch_cfg<1, 2, 1, 1>(a, stream);
  else if (tn == 128 && tk == 32) rc = launch_cfg<4, 1, 1, 1>(a, stream);
  else if (tn == 64 && tk == 32) rc = launch_cfg<2, 1, 1, 1>(a, stream);
  else rc = launch_cfg<1, 1, 1, 1>(a, stream);
  return rc;
}

int kodhip_conv_wgrad(const void* x, const void* dy, float* partials, float* grad,
                      int B, int H, int W, int ldx, int xcoff, int Cin,
                      int N, int KH, int KW, int SH, int SW, int PH, int PW, int Kp,
                      int ldy, int ycoff, int n_valid, int stem, float scale, hipStream_t stream) {
  KOD_CHECK_ARG(grad, "conv_wgrad: null pointer");
  KOD_CHECK_ARG(n_valid > 0 && n_valid <= N, "conv_wgrad: bad n_valid");
  WgradArgs a;
  if (int rc = wgrad_partial(a, x, dy, partials, B, H, W, ldx, xcoff, Cin, N, KH, KW, SH, SW, PH, PW, Kp, ldy, ycoff, stream)) return rc;
  return launch_wgrad_reduce((const float*)partials, grad, a.splits, N, n_valid, a.K, Kp, stem ? 8 : Cin, KH * KW, stem, scale,
                             nullptr, 1 << 30, stream);
}

// The split-K half alone: fp32 slabs partials[kodhip_conv_wgrad_splits(M, N, Kp)][N][Kp]; kodhip_wgrad_reduce_batched
// turns the slabs of many layers into gradients with one launch.
int kodhip_conv_wgrad_partial(const void* x, const void* dy, float* partials,
                              int B, int H, int W, int ldx, int xcoff, int Cin,
                              int N, int KH, int KW, int SH, int SW, int PH, int PW, int Kp,
                              int ldy, int ycoff, hipStream_t stream) {
  WgradArgs a;
  return wgrad_partial(a, x, dy, partials, B, H, W, ldx, xcoff, Cin, N, KH, KW, SH, SW, PH, PW, Kp, ldy, ycoff, stream);
}

// Weight gradients of TWO pointwise (1x1 / stride 1) layers that read the same input - a CSP layer's main_conv and
// short_conv (kod/nn/layers/csp.py:85-99) - in one launch + one reduction: N slab rows per layer, the second layer's n tiles
// take dY from dy2.  Blocks of one split share an XCD, so the input tile a split streams is fetched from HBM once for
// both layers (these launches are HBM-bound: 64 -> 32 @160 moves 315 MB per layer, 210 MB of it the shared input).
// kodhip_conv_wgrad_dual_splits: slab count (partials: splits * 2N * Kp floats), 0 when the form does not apply (operands
// beyond the 32-bit buffer range, KODHIP_WGRAD_DMA=none) - the caller then launches kodhip_conv_wgrad twice.
static bool wgrad_dual_ok(long M, int B, int H, int W, int ldx, int ldy) {
  static const char* mode = getenv("KODHIP_WGRAD_DMA");
  const long xb = (long)B * H * W * ldx * 2, yb = M * ldy * 2;
  return !(mode && mode[0] == 'n') && xb < (1l << 32) - 64 && yb < (1l << 32) - 64;
}

static int wgrad_rows_per_split_dual(long M, int N, int Kp) {
  int tn, tk;
  tile_shape(N, Kp, &tn, &tk);
  const int tiles = 2 * cdiv(N, tn) * cdiv(Kp, tk);
  static int slots = 0;
  if (!slots) { const char* e = getenv("KODHIP_WGRAD_SLOTS"); slots = e ? atoi(e) : 512; if (slots < 8) slots = 512; }
  int s = slots / tiles;
  if (s < 1) s = 1;
  const long maxs = (M + 255) / 256;
  if (s > maxs) s = (int)maxs;
  if (s < 1) s = 1;
  return cdiv(cdiv(M, s), 32) * 32;
}

int kodhip_conv_wgrad_dual_splits(int B, int H, int W, int ldx, int Cin, int N, int Kp, int ldy) {
  (void)Cin;
  const long M = (long)B * H * W;
  if (!wgrad_dual_ok(M, B, H, W, ldx, ldy)) return 0;
  return (int)cdiv(M, (long)wgrad_rows_per_split_dual(M, N, Kp));
}

int kodhip_conv_wgrad_dual(const void* x, const void* dy1, const void* dy2, float* partials, float* grad1, float* grad2,
                           int B, int H, int W, int ldx, int xcoff, int Cin, int N, int Kp, int ldy, int ycoff,
                           float scale, hipStream_t stream) {
  KOD_CHECK_ARG(x && dy1 && dy2 && partials && grad1 && grad2, "conv_wgrad_dual: null pointer");
  KOD_CHECK_ARG(Cin % 8 == 0 && ldx % 8 == 0 && xcoff % 8 == 0 && xcoff + Cin <= ldx, "conv_wgrad_dual: bad input slice");
  KOD_CHECK_ARG(N % 8 == 0 && ldy % 8 == 0 && ycoff % 8 == 0 && ycoff + N <= ldy, "conv_wgrad_dual: bad dy slice (N=%d ldy=%d)", N, ldy);
  KOD_CHECK_ARG(Kp % 32 == 0 && Kp >= Cin, "conv_wgrad_dual: bad Kp");
  const long M = (long)B * H * W;
  KOD_CHECK_ARG(M < (1l << 31), "conv_wgrad_dual: pixel count overflows int32");
  KOD_CHECK_ARG(wgrad_dual_ok(M, B, H, W, ldx, ldy), "conv_wgrad_dual: not available for this geometry (kodhip_conv_wgrad_dual_splits == 0)");
  WgradArgs a = {};
  a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dy1; a.dy2 = (const bf16_t*)dy2; a.part = partials;
  a.B = B; a.Hs = H; a.Ws = W; a.ldx = ldx; a.xcoff = xcoff; a.Cin = Cin;
  a.Ho = H; a.Wo = W; a.M = (int)M; a.N = 2 * N; a.n_half = N; a.K = Cin; a.Kp = Kp;
  a.KH = a.KW = a.SH = a.SW = 1; a.PH = a.PW = 0; a.ldy = ldy; a.ycoff = ycoff;
  a.magic_cin = magic_u32((uint32_t)Cin); a.magic_kw = magic_u32(1u);
  a.magic_hwo = magic_u32((uint32_t)(H * W)); a.magic_wo = magic_u32((uint32_t)W);
  a.m_per_split = wgrad_rows_per_split_dual(M, N, Kp);
  a.splits = cdiv(M, a.m_per_split);
  int tn, tk, rc;
  tile_shape(N, Kp, &tn, &tk);
  if (tn == 128 && tk == 128) rc = launch_cfg<2, 2, 2, 2>(a, stream);
  else if (tn == 64 && tk == 128) rc = launch_cfg<2, 2, 1, 2>(a, stream);
  else if (tn == 32 && tk == 128) rc = launch_cfg<1, 4, 1, 1>(a, stream);
  else if (tn == 128 && tk == 64) rc = launch_cfg<2, 2, 2, 1>(a, stream);
  else if (tn == 64 && tk == 64) rc = launch_cfg<2, 2, 1, 1>(a, stream);
  else if (tn == 32 && tk == 64) rc = launch_cfg<1, 2, 1, 1>(a, stream);
  else if (tn == 128 && tk == 32) rc = launch_cfg<4, 1, 1, 1>(a, stream);
  else if (tn == 64 && tk == 32) rc = launch_cfg<2, 1, 1, 1>(a, stream);
  else rc = launch_cfg<1, 1, 1, 1>(a, stream);
  if (rc) return rc;
  return launch_wgrad_reduce((const float*)partials, grad1, a.splits, 2 * N, 2 * N, a.K, Kp, Cin, 1, 0, scale, grad2, N, stream);
}

// The stem's BatchNorm/SiLU backward + weight gradient as one kernel (conv_stem_bwd_fused_kernel) followed by the slab
// reduction: replaces kodhip_bn_silu_bwd_apply + kodhip_conv_wgrad(stem = 1) for the unit that has no data gradient
// (kod/nn/backbones/yolov5.py:44-52: the 6x6 / stride 2 / pad 2 stem; aten::native_batch_norm_backward + silu_backward +
// convolution_backward dW).  x: pixel pairs [B][H][Wp][8] bf16 (Wp = image width / 2), dA / y: [B * H/2 * Wp][N] slices,
// coef = kodhip_bn_bwd_coeffs*'s k1 | k2 | k3; partials: kodhip_stem_bwd_fused_blocks(B, H, Wp, N) * (N <= 32 ? 32 : 64) * 160
// floats; grad: fp32 [N][3][6][6].  y is left untouched (dY is never materialised).  N <= 64.
// tile width: 80 pixels (four 5-wave blocks per CU; measured 336 us at B = 64 / 640 px against 443 us for 160-pixel tiles
// with two blocks per CU); KODHIP_STEM_BWD_TW = 80 | 160 is the A/B knob
static int stem_bwd_tw(int Wp) {
  static int tw = 0;
  if (!tw) { const char* e = getenv("KODHIP_STEM_BWD_TW"); tw = e ? atoi(e) : 80; if (tw != 80 && tw != 160) tw = 80; }
  return Wp <= 80 ? 80 : tw;
}

int kodhip_stem_bwd_fused_blocks(int B, int H, int Wp, int N) {
  const int TW = N > 32 ? 80 : stem_bwd_tw(Wp);
  const long tiles = (long)B * (H / 2) * cdiv(Wp, TW);
  static int slots = 0;                  // KODHIP_STEM_BWD_BLOCKS: A/B knob (default: every resident slot of the chip)
  if (!slots) { const char* e = getenv("KODHIP_STEM_BWD_BLOCKS"); slots = e ? atoi(e) : 0; if (slots < 1) slots = 0; }
  const int want = slots ? slots : (N > 32 ? 512 : (TW == 160 ? 512 : 1024));      // resident blocks: 2 / 2 / 4 per CU
  const int tpb = cdiv(tiles, want);
  return cdiv(tiles, tpb);
}

int kodhip_stem_bwd_fused(const void* x, const void* dA, int lda, int dacoff, const void* y, int ldy,
                          const float* scale, const float* shift, const float* coef, float* partials, float* grad,
                          int B, int H, int Wp, int N, float gscale, hipStream_t stream) {
  KOD_CHECK_ARG(x && dA && y && scale && shift && coef && partials && grad, "stem_bwd_fused: null pointer");
  KOD_CHECK_ARG(B > 0 && H >= 2 && H % 2 == 0 && Wp > 0 && N > 0 && N <= 64 && N % 8 == 0, "stem_bwd_fused: bad geometry");
  KOD_CHECK_ARG(lda % 8 == 0 && dacoff % 8 == 0 && dacoff + N <= lda && ldy % 8 == 0 && ldy >= N, "stem_bwd_fused: bad slices");
  StemBwdArgs a = {};
  a.x = (const bf16_t*)x; a.dA = (const bf16_t*)dA; a.lda = lda; a.dacoff = dacoff; a.y = (const bf16_t*)y; a.ldy = ldy;
  a.scale = scale; a.shift = shift; a.coef = coef; a.part = partials;
  a.B = B; a.Hs = H; a.Wp = Wp; a.Ho = H / 2; a.Wo = Wp; a.N = N;
  const long M = (long)B * a.Ho * a.Wo;
  const long xb = (long)B * H * Wp * 16, gb = M * lda * 2, yb = M * ldy * 2;
  KOD_CHECK_ARG(xb < (1l << 32) - 64 && gb < (1l << 32) - 64 && yb < (1l << 32) - 64, "stem_bwd_fused: tensor beyond the 32-bit buffer range");
  a.x_bytes = (uint32_t)xb; a.da_bytes = (uint32_t)gb; a.y_bytes = (uint32_t)yb;
  const int TW = N > 32 ? 80 : stem_bwd_tw(Wp);                 // (the two-tile form exists for 80-pixel tiles only)
  a.tiles_per_row = cdiv(a.Wo, TW);
  a.tiles = B * a.Ho * a.tiles_per_row;
  const int blocks = kodhip_stem_bwd_fused_blocks(B, H, Wp, N);
  a.tiles_per_block = cdiv(a.tiles, blocks);
  if (N > 32) hipLaunchKernelGGL((conv_stem_bwd_fused_kernel<80, 2>), dim3(blocks), dim3(320), 0, stream, a);
  else if (TW == 160) hipLaunchKernelGGL((conv_stem_bwd_fused_kernel<160, 1>), dim3(blocks), dim3(320), 0, stream, a);
  else hipLaunchKernelGGL((conv_stem_bwd_fused_kernel<80, 1>), dim3(blocks), dim3(320), 0, stream, a);
  KOD_LAUNCH_CHECK("stem_bwd_fused");
  return launch_wgrad_reduce((const float*)partials, grad, blocks, N > 32 ? 64 : 32, N, 144, 160, 8, 18, 1, gscale, nullptr, 1 << 30, stream);
}

int kodhip_wgrad_reduce_desc_bytes(void) { return (int)sizeof(ReduceDesc); }
int kodhip_wgrad_reduce_blocks(int n_valid, int K) { return cdiv(n_valid * K, RED_K); }

// descs: DEVICE array of n_desc KodWgradReduceDesc sorted by block_start (block_start[0] = 0, each layer owning
// kodhip_wgrad_reduce_blocks(N, K) consecutive blocks, total_blocks in all); offsets relative to partials / grads.
int kodhip_wgrad_reduce_batched(const float* partials, float* grads, const void* descs, int n_desc, int total_blocks,
                                hipStream_t stream) {
  KOD_CHECK_ARG(partials && grads && descs && n_desc > 0 && total_blocks > 0, "wgrad_reduce_batched: bad args");
  hipLaunchKernelGGL(wgrad_reduce_batched_kernel, dim3(total_blocks), dim3(256), 0, stream, partials, grads,
                     (const ReduceDesc*)descs, n_desc);
  KOD_LAUNCH_CHECK("wgrad_reduce_batched");
  return KOD_OK;
}

}  // extern "C"
