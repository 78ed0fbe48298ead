// Small HBM-bound helpers of the YOLOv5 train step (gfx950):
//   * NCHW fp32 image -> channels-last bf16 (C padded to 4, so the 6x6/s2 stem becomes a 6x3 conv over
//     8-channel pixel pairs with 16-byte k-runs)
//   * fp32 master weights [Cout][Cin][KH][KW] -> bf16 MFMA packs (forward / wgrad order and dgrad order)
//   * SPPF 5x5/s1/p2 max-pool forward (+argmax) and backward (kod/nn/layers/sppf.py:46-50,73-76)
//   * nearest x2 upsample forward / backward (kod/nn/necks/yolov5_pafpn.py:144-146,182-184)
//   * detection-head gradient re-layout [B,A,h,w,5+nc] fp32 -> [M][Npad] bf16 + bias gradient
//   * fused multi-tensor Nesterov SGD (torch.optim.SGD as configured by kod/nn/optim/smart.py:36-58)
#include "kodhip_common.h"

namespace {

__global__ void nchw_to_nhwc4_kernel(const float* x, bf16_t* y, int B, int C, long HW) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * HW) return;
  long b = i / HW, p = i - b * HW;
  bf16x4 v;
#pragma unroll
  for (int c = 0; c < 4; ++c) v[c] = (bf16_t)(c < C ? x[(b * C + c) * HW + p] : 0.f);
  *reinterpret_cast<bf16x4*>(y + i * 4) = v;
}

// HW % 4 == 0 (every training geometry): 4 consecutive pixels per thread - one 16-byte load per plane, one 32-byte
// store - so that each lane has 48 + 32 bytes in flight instead of 12 + 8 (the pass is the first kernel of the step
// and nothing overlaps it: 3.9 -> 5 TB/s)
__global__ void nchw_to_nhwc4_x4_kernel(const float* x, bf16_t* y, int B, int C, long HW) {
  const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;          // pixel quad
  const long HWq = HW >> 2;
  if (q >= (long)B * HWq) return;
  const long b = q / HWq, pq = q - b * HWq;
  f32x4 pl[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
    pl[c] = c < C ? kod_load_once<f32x4>(x + (b * C + c) * HW + pq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 o0, o1;
  const float* f0 = reinterpret_cast<const float*>(&pl[0]);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    o0[c] = (bf16_t)f0[c * 4 + 0]; o0[4 + c] = (bf16_t)f0[c * 4 + 1];
    o1[c] = (bf16_t)f0[c * 4 + 2]; o1[4 + c] = (bf16_t)f0[c * 4 + 3];
  }
  bf16_t* dst = y + (b * HW + pq * 4) * 4;
  *reinterpret_cast<bf16x8*>(dst) = o0;
  *reinterpret_cast<bf16x8*>(dst + 8) = o1;
}

struct PackDesc {          // all int64 so the host can fill it as a plain int64[13] row
  long w_off;              // offset (elements) of this weight in the fp32 master arena
  long f_off;              // offset of its first row in the bf16 forward pack arena ([N][Kp] rows)
  long d_off;              // offset of the layer's dgrad pack ([Cin][Kdp]), -1: none
  long N, Cin, KH, KW;     // true weight dims [N][Cin][KH][KW]
  long Kp, Kdp;            // padded row lengths of the two packs
  long Ntot, n_off;        // dgrad pack: channels per tap (padded total) and this weight's first channel
  long stem;               // 1: 6x6/s2 stem in pixel-pair form, k = kh*32 + kw'*8 + dx*4 + c (kw' = 0..2 of a 4-pair,
                           //    32-value window: the 4th pair's weights stay zero); 2: 3x3/s2 parity-class dgrad packs;
                           //    3: 3x3/s2 folded dgrad pack (kodhip_conv_dgrad_s2f)
  long blk_begin;          // first block of this descriptor in the grid
};

// One launch packs every layer: block -> descriptor by binary search on blk_begin.  Only valid entries
// are written; padding stays zero from the one-time arena memset.
__global__ void pack_weights_kernel(const float* master, bf16_t* fpack, bf16_t* dpack, const PackDesc* descs,
                                    int nlayers) {
  int lo = 0, hi = nlayers - 1;
  long blk = blockIdx.x;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (descs[mid].blk_begin <= blk) lo = mid; else hi = mid - 1;
  }
  const PackDesc d = descs[lo];
  long local = (blk - d.blk_begin) * blockDim.x + threadIdx.x;
  const long KK = d.KH * d.KW;
  const long count = d.N * d.Cin * KK;          // true element count
  if (local >= count) return;
  // local enumerates the master layout [n][ci][tap] so reads are coalesced
  long n = local / (d.Cin * KK);
  long rem = local - n * d.Cin * KK;
  long ci = rem / KK;
  long tap = rem - ci * KK;
  bf16_t v = (bf16_t)master[d.w_off + local];
  if (d.stem == 1) {
    long kh = tap / 6, kw = tap - kh * 6;
    long k = kh * 32 + (kw >> 1) * 8 + (kw & 1) * 4 + ci;
    fpack[d.f_off + n * d.Kp + k] = v;
  } else {
    // packed K axes are tap-major with each tap padded to a multiple of 32 channels (zeros): k = tap * round_up(C, 32) + c,
    // so that a 32-wide MFMA K step never straddles two taps whatever the channel count (yv5m: 48)
    const long cs = (d.Cin + 31) / 32 * 32;
    const long ns = (d.Ntot + 31) / 32 * 32;
    fpack[d.f_off + n * d.Kp + tap * cs + ci] = v;
    if (d.d_off >= 0) {
      if (d.stem == 2) {
        // 3x3 stride-2 layer: four parity-class packs (see kodhip_conv_dgrad_s2)
        long kh = tap / 3, kw = tap - kh * 3;
        long py = kh != 1, px = kw != 1;
        long khp = kh == 2, kwp = kw == 2;
        long base = d.d_off;
        for (long c = 0; c < 2 * py + px; ++c) {
          long nt = (1 + (c >> 1)) * (1 + (c & 1));
          base += d.Cin * (nt * ns);
        }
        long KW = 1 + px;
        long Kc = (1 + py) * KW * ns;
        dpack[base + ci * Kc + (khp * KW + kwp) * ns + n] = v;
      } else if (d.stem == 3) {
        // 3x3 stride-2 layer, folded form: row = class * Cin + ci, k = (dy * 2 + dx) * ns + n over the 2x2 dY neighbourhood
        long kh = tap / 3, kw = tap - kh * 3;
        long cls = 2 * (kh != 1) + (kw != 1);
        long t2 = 2 * (kh == 0) + (kw == 0);
        dpack[d.d_off + (cls * d.Cin + ci) * (4 * ns) + t2 * ns + n] = v;
      } else {
        dpack[d.d_off + ci * d.Kdp + tap * ns + d.n_off + n] = v;
      }
    }
  }
}

// ------------------------------------------------------------------ SPPF max pool 5x5 s1 p2
// These kernels are bound by vector-instruction issue, not by memory (13 MB per launch at 20 x 20: a wave64 instruction
// takes its 16-lane SIMD four cycles, and comparing every (element, tap) pair in fp32 with torch's NaN rule cost ~6.4
// instructions per pair: 41 us).  The forward kernel therefore works on integer KEYS:
//   K = sortable(value) << 16 | (127 - (16 * row + column) in the thread's 8 x 8 input patch)   (0 for taps outside the image)
// sortable(): bf16 bits h >= 0 -> h | 0x8000, h < 0 -> ~h, NaN -> 0xffff; an unsigned maximum over the window's keys is
// then torch's scan (kod/nn/layers/sppf.py:46-50 -> F.max_pool2d: row-major over the window, the FIRST maximum wins - the
// position field breaks ties towards the earlier tap - and a NaN beats every number).  One thread owns a 4 x 4 block of
// outputs x 4 channels: an 8 x 8 patch is encoded once (not once per output row), the row maxima of a patch row are
// shared by the four outputs of that row (10 instead of 16 maxima), the column pass works the same way, and value and
// argmax fall out of the winning key.  Deviations from torch, all without numerical consequence: a window whose maximum
// is a tie between +0 and -0 yields +0 (torch: whichever comes first); of several NaNs in a window the first is
// reported (torch: the last); a NaN leaves as 0x7fff.
// idx byte = dy * 16 + dx of the winning tap (dy, dx in 0..4), read by maxpool5_bwd_kernel.
__device__ __forceinline__ uint32_t pool_key(const uint32_t x /* fp32 bits of the bf16 value */, const uint32_t code) {
  // x >= 0: flip the top bit; x < 0: complement (arithmetic, no condition registers: the kernel is issue-bound)
  int32_t sg = (int32_t)x >> 31;
  asm volatile("" : "+v"(sg));                              // (or the compiler turns it back into compare + select)
  const uint32_t t = (uint32_t)sg | 0x80000000u;
  const uint32_t k = ((x ^ t) & 0xffff0000u) | code;
  const float f = __uint_as_float(x);
  return (f != f) ? (0xffff0000u | code) : k;
}

// window maxima of 5 over 8 keys -> 4 outputs (o = 0..3 covers k[o .. o + 4]), 10 maxima
__device__ __forceinline__ void pool_max5of8(const uint32_t (&k)[8], uint32_t (&out)[4]) {
  const uint32_t m34 = max(k[3], k[4]);
  const uint32_t m234 = max(m34, k[2]), m345 = max(m34, k[5]);
  const uint32_t m2345 = max(m234, k[5]);
  out[0] = max(max(k[0], k[1]), m234);
  out[1] = max(k[1], m2345);
  out[2] = max(k[6], m2345);
  out[3] = max(max(k[6], k[7]), m345);
}

// one thread's 4 x 4 outputs x 4 channels: pixel group (gy, gx) of image b, channel quad c4
__device__ __forceinline__ void maxpool5_fwd_body(const bf16_t* x, int ldx, int xcoff, bf16_t* y, int ldy, int ycoff,
                                                  unsigned char* idx, int H, int W, int C, int b, int gy, int gx, int c4) {
  const int ox0 = gx * 4, oy0 = gy * 4;
  // patch column c <-> ix = ox0 - 2 + c, patch row r <-> iy = oy0 - 2 + r: clamped addresses, all-ones / zero masks
  uint32_t cmask[8], rmask[8];
  int cx[8], ry[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int ix = ox0 - 2 + c, iy = oy0 - 2 + c;
    cmask[c] = (ix >= 0 && ix < W) ? 0xffffffffu : 0u;
    rmask[c] = (iy >= 0 && iy < H) ? 0xffffffffu : 0u;
    cx[c] = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
    ry[c] = iy < 0 ? 0 : (iy >= H ? H - 1 : iy);
    asm volatile("" : "+v"(cmask[c]), "+v"(rmask[c]));      // plain AND masks (not selects on a condition register pair)
  }
  const bf16_t* xb = x + (long)b * H * W * ldx + xcoff + c4 * 4;
  uint32_t hm[8][4][4];            // [patch row][output column][channel]: that row's maximum over the output's 5 columns
  uint2 cur[8], nxt[8];
  auto load_row = [&](const int r, uint2 (&dst)[8]) {
    const bf16_t* row = xb + (long)ry[r] * W * ldx;
#pragma unroll
    for (int c = 0; c < 8; ++c) dst[c] = *reinterpret_cast<const uint2*>(row + (long)cx[c] * ldx);
  };
  load_row(0, cur);
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    if (r + 1 < 8) load_row(r + 1, nxt);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      uint32_t k[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const uint32_t d = (e < 2) ? cur[c].x : cur[c].y;
        const uint32_t xf = (e & 1) ? (d & 0xffff0000u) : (d << 16);
        k[c] = pool_key(xf, (uint32_t)(127 - (r * 16 + c))) & cmask[c];
      }
      uint32_t o4[4];
      pool_max5of8(k, o4);
#pragma unroll
      for (int o = 0; o < 4; ++o) hm[r][o][e] = o4[o] & rmask[r];
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) cur[c] = nxt[c];
  }
  // column pass + decode + store
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    uint32_t best[4][4];             // [output row][channel]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      uint32_t k[8], o4[4];
#pragma unroll
      for (int r = 0; r < 8; ++r) k[r] = hm[r][o][e];
      pool_max5of8(k, o4);
#pragma unroll
      for (int q = 0; q < 4; ++q) best[q][e] = o4[q];
    }
    if (ox0 + o >= W) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (oy0 + q >= H) continue;
      uint32_t hv[4], ib = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t K = best[q][e];
        // upper half back to bf16 bits: K >= 2^31 (h >= 0, or NaN -> 0x7fff): flip the top bit; else complement
        const uint32_t t = (uint32_t)((int32_t)K >> 31) & 0x7fffffffu;
        hv[e] = ~(K ^ t);
        // position -> tap: patch position = 127 - (K & 127); minus the window's origin (q, o) = dy * 16 + dx
        ib |= ((uint32_t)(127 - (q * 16 + o)) - (K & 127u)) << (8 * e);
      }
      const long op = (long)(b * H + oy0 + q) * W + ox0 + o;
      uint2 v;
      v.x = (hv[0] >> 16) | (hv[1] & 0xffff0000u);
      v.y = (hv[2] >> 16) | (hv[3] & 0xffff0000u);
      *reinterpret_cast<uint2*>(y + op * ldy + ycoff + c4 * 4) = v;
      *reinterpret_cast<uint32_t*>(idx + op * C + c4 * 4) = ib;
    }
  }
}

__global__ __launch_bounds__(256) void maxpool5_fwd_kernel(const bf16_t* x, int ldx, int xcoff, bf16_t* y, int ldy, int ycoff,
                                    unsigned char* idx, int B, int H, int W, int C) {
  const int C4 = C >> 2;
  const int WG = (W + 3) >> 2, HG = (H + 3) >> 2;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * HG * WG * C4;
  if (i >= total) return;
  const int c4 = (int)(i % C4);
  long p = i / C4;
  const int gx = (int)(p % WG);
  p /= WG;
  const int gy = (int)(p % HG);
  const int b = (int)(p / HG);
  maxpool5_fwd_body(x, ldx, xcoff, y, ldy, ycoff, idx, H, W, C, b, gy, gx, c4);
}

// dx[p] += sum over outputs q whose argmax is p of dy[q]   (gather form, deterministic).  Issue-bound like the forward
// kernel: comparing every (input, output) pair in registers costs ~6 instructions per pair, 25 pairs per input element
// (34 us).  Here a thread owns a 4 x 4 block of input pixels x 4 channels and keeps their 64 sums in PRIVATE LDS words
// (word s of thread t at [s][t]: a thread's words never leave its bank column, so a wave's scattered accesses are
// conflict-free; one array per channel, so that the compiler knows the channels' chains do not alias); it walks the
// 8 x 8 outputs whose windows reach the block ONCE, in the fixed order (bottom output row first, left to right), turns
// each stored tap byte into the block position it points at -
//   t = byte + 16 r + c - 0x44 = 16 (r + dy - 4) + (c + dx - 4), inside the block iff (t & ~0x33) == 0
// (0x1000 added for outputs outside the image) - and adds the gradient to that word, or to a dump word: one LDS
// read-modify-write per OUTPUT element, no per-pair compares (84 instead of 156 instructions per input element: 27 us;
// the LDS float atomic, ds_add_f32, does the same in 176 us).  A thread's accesses to its words execute in program
// order, so the sums are those of the register form this replaces, in the same order.
constexpr int POOL_BWD_THREADS = 128;       // 35 KB of private words per block: four blocks per CU
// one thread's 4 x 4 inputs x 4 channels: pixel group (gy, gx) of image b, channel quad c4; words0..3: this thread's column
// of the block's private-word arrays (stride POOL_BWD_THREADS floats between a thread's words)
template <int POOL_BWD_THREADS>
__device__ __forceinline__ void maxpool5_bwd_body(float* w0, float* w1, float* w2, float* w3,
                                                  const bf16_t* dy, int ldy, int ycoff, const unsigned char* idx,
                                                  bf16_t* dx, int ldx, int xcoff, int H, int W, int C, const float* dx32,
                                                  int b, int gy, int gx, int c4) {
  const int ix0 = gx * 4, iy0 = gy * 4;
  float* const mine4[4] = {w0, w1, w2, w3};
#pragma unroll
  for (int ri = 0; ri < 4; ++ri)
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      const bool in = iy0 + ri < H && ix0 + ci < W;
      const long po = ((long)(b * H + (in ? iy0 + ri : 0)) * W + (in ? ix0 + ci : 0)) * ldx + xcoff + c4 * 4;
      uint2 old = {0u, 0u};
      if (in && !dx32) old = *reinterpret_cast<const uint2*>(dx + po);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t d = e < 2 ? old.x : old.y;
        mine4[e][(ri * 4 + ci) * POOL_BWD_THREADS] =
            in ? (dx32 ? dx32[po + e] : __uint_as_float((e & 1) ? (d & 0xffff0000u) : (d << 16))) : 0.f;   // dx32: fp32 partial
      }
    }
  // patch column c <-> ox = ix0 - 2 + c, patch row r <-> oy = iy0 - 2 + r: clamped addresses, 0x1000 for outside
  uint32_t ck[8], rk[8];
  int coff_i[8], coff_g[8];
  long roff[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int ox = ix0 - 2 + c, oy = iy0 - 2 + c;
    ck[c] = (uint32_t)c + ((ox >= 0 && ox < W) ? 0u : 0x1000u);
    rk[c] = (uint32_t)(16 * c - 0x44) + ((oy >= 0 && oy < H) ? 0u : 0x1000u);
    const int cxc = ox < 0 ? 0 : (ox >= W ? W - 1 : ox), ryc = oy < 0 ? 0 : (oy >= H ? H - 1 : oy);
    coff_i[c] = cxc * C;
    coff_g[c] = cxc * ldy;
    roff[c] = (long)(b * H + ryc) * W;
  }
  const unsigned char* ibase = idx + c4 * 4;
  const bf16_t* gbase = dy + ycoff + c4 * 4;
  uint32_t ibc[8], ibn[8];
  uint2 gc[8], gn[8];
  auto load_row = [&](const int r, uint32_t (&ib)[8], uint2 (&g)[8]) {
    const unsigned char* ir = ibase + roff[r] * C;
    const bf16_t* gr = gbase + roff[r] * ldy;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      ib[c] = *reinterpret_cast<const uint32_t*>(ir + coff_i[c]);           // 4 argmax bytes in one load
      g[c] = *reinterpret_cast<const uint2*>(gr + coff_g[c]);
    }
  };
  load_row(7, ibc, gc);
#pragma unroll
  for (int r = 7; r >= 0; --r) {
    if (r > 0) load_row(r - 1, ibn, gn);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      uint32_t word[4];
      float g[4], cur[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t t = ((ibc[c] >> (8 * e)) & 0xffu) + rk[r] + ck[c];
        const uint32_t pos = ((t >> 2) & 12u) | (t & 3u);
        word[e] = (t & ~0x33u) == 0u ? pos : 16u;
        const uint32_t d = e < 2 ? gc[c].x : gc[c].y;
        g[e] = __uint_as_float((e & 1) ? (d & 0xffff0000u) : (d << 16));
      }
      // this thread's own words (LDS float atomics measured 5 x slower); the four channels' reads leave together
#pragma unroll
      for (int e = 0; e < 4; ++e) cur[e] = mine4[e][word[e] * POOL_BWD_THREADS];
#pragma unroll
      for (int e = 0; e < 4; ++e) mine4[e][word[e] * POOL_BWD_THREADS] = cur[e] + g[e];
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) { ibc[c] = ibn[c]; gc[c] = gn[c]; }
  }
#pragma unroll
  for (int ri = 0; ri < 4; ++ri) {
    if (iy0 + ri >= H) continue;
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      if (ix0 + ci >= W) continue;
      bf16x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (bf16_t)mine4[e][(ri * 4 + ci) * POOL_BWD_THREADS];
      *reinterpret_cast<bf16x4*>(dx + ((long)(b * H + iy0 + ri) * W + ix0 + ci) * ldx + xcoff + c4 * 4) = v;
    }
  }
}

__global__ __launch_bounds__(POOL_BWD_THREADS) void maxpool5_bwd_kernel(const bf16_t* dy, int ldy, int ycoff, const unsigned char* idx,
                                    bf16_t* dx, int ldx, int xcoff, int B, int H, int W, int C, const float* dx32) {
  // one array per channel: the compiler then knows that the four channels' read-modify-write chains do not alias
  __shared__ float words0[17 * POOL_BWD_THREADS], words1[17 * POOL_BWD_THREADS], words2[17 * POOL_BWD_THREADS], words3[17 * POOL_BWD_THREADS];      // [block position 0..15 + dump][thread]
  const int C4 = C >> 2;
  const int WG = (W + 3) >> 2, HG = (H + 3) >> 2;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * HG * WG * C4;
  if (i >= total) return;
  const int c4 = (int)(i % C4);
  long p = i / C4;
  const int gx = (int)(p % WG);
  p /= WG;
  const int gy = (int)(p % HG);
  const int b = (int)(p / HG);
  maxpool5_bwd_body<POOL_BWD_THREADS>(words0 + threadIdx.x, words1 + threadIdx.x, words2 + threadIdx.x, words3 + threadIdx.x,
                                      dy, ldy, ycoff, idx, dx, ldx, xcoff, H, W, C, dx32, b, gy, gx, c4);
}

// ------------------------------------------------------------------ max-pool of any odd window (SPPF kernel_sizes != 5)
// SPPFBottleneck takes any kernel size or sequence of sizes (kod/nn/layers/sppf.py:27-67); the network's 5 (and the SPP
// sequence 5 / 9 / 13 as its cascade) runs through maxpool5_* above, every other window through these two plain kernels -
// one thread per (pixel, 8 channels), torch's scan: row-major over the window, `val > max || isnan(val)` replaces (so the
// FIRST maximum and the LAST NaN win, aten/native/cpu/MaxPoolKernel.cpp), -inf padding.  idx byte = dy * 16 + dx (K <= 15).
__global__ __launch_bounds__(256) void maxpool_k_fwd_kernel(const bf16_t* x, int ldx, int xcoff, bf16_t* y, int ldy, int ycoff,
                                                            unsigned char* idx, int B, int H, int W, int C, int K) {
  const int C8 = C >> 3;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * H * W * C8) return;
  const int c8 = (int)(i % C8);
  const long p = i / C8;
  const int ox = (int)(p % W), oy = (int)((p / W) % H), b = (int)(p / ((long)W * H));
  const int r = K >> 1;
  float best[8];
  int tap[8];
  bool first = true;
  for (int dy = 0; dy < K; ++dy) {
    const int iy = oy + dy - r;
    if (iy < 0 || iy >= H) continue;
    for (int dx = 0; dx < K; ++dx) {
      const int ix = ox + dx - r;
      if (ix < 0 || ix >= W) continue;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + ((long)(b * H + iy) * W + ix) * ldx + xcoff + c8 * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float f = (float)v[e];
        if (first) { best[e] = -INFINITY; tap[e] = dy * 16 + dx; }
        if (f > best[e] || f != f) { best[e] = f; tap[e] = dy * 16 + dx; }
      }
      first = false;
    }
  }
  bf16x8 o;
  unsigned long long ib = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) { o[e] = (bf16_t)best[e]; ib |= (unsigned long long)(tap[e] & 0xff) << (8 * e); }
  *reinterpret_cast<bf16x8*>(y + p * ldy + ycoff + c8 * 8) = o;
  *reinterpret_cast<unsigned long long*>(idx + p * C + c8 * 8) = ib;
}

// dx[p] += sum over the outputs whose argmax is p (gather form: fixed order, deterministic); dx32 as in maxpool5_bwd_kernel
__global__ __launch_bounds__(256) void maxpool_k_bwd_kernel(const bf16_t* dy, int ldy, int ycoff, const unsigned char* idx,
                                                            bf16_t* dx, int ldx, int xcoff, int B, int H, int W, int C, int K,
                                                            const float* dx32) {
  const int C8 = C >> 3;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * H * W * C8) return;
  const int c8 = (int)(i % C8);
  const long p = i / C8;
  const int ix = (int)(p % W), iy = (int)((p / W) % H), b = (int)(p / ((long)W * H));
  const int r = K >> 1;
  float acc[8];
  bf16_t* d = dx + p * ldx + xcoff + c8 * 8;
  if (dx32) {
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = dx32[p * ldx + xcoff + c8 * 8 + e];
  } else {
    const bf16x8 old = *reinterpret_cast<const bf16x8*>(d);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = (float)old[e];
  }
  for (int oy = max(iy - r, 0); oy <= min(iy + r, H - 1); ++oy)
    for (int ox = max(ix - r, 0); ox <= min(ix + r, W - 1); ++ox) {
      const long q = (long)(b * H + oy) * W + ox;
      const unsigned long long ib = *reinterpret_cast<const unsigned long long*>(idx + q * C + c8 * 8);
      const bf16x8 g = *reinterpret_cast<const bf16x8*>(dy + q * ldy + ycoff + c8 * 8);
      const unsigned me = (unsigned)((iy - oy + r) * 16 + (ix - ox + r));
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (((ib >> (8 * e)) & 0xffu) == me) acc[e] += (float)g[e];
    }
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (bf16_t)acc[e];
  *reinterpret_cast<bf16x8*>(d) = o;
}

// ------------------------------------------------------------------ nearest x2 upsample
__global__ void upsample2x_fwd_kernel(const bf16_t* x, int ldx, int xcoff, bf16_t* y, int ldy, int ycoff,
                                      int B, int H, int W, int C) {   // H,W = input dims
  const int CC = C >> 3;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)B * 4 * H * W * CC;
  if (i >= total) return;
  int cc = (int)(i % CC);
  long p = i / CC;
  int ox = (int)(p % (2 * W));
  long q = p / (2 * W);
  int oy = (int)(q % (2 * H));
  int b = (int)(q / (2 * H));
  bf16x8 v = *reinterpret_cast<const bf16x8*>(x + ((long)(b * H + (oy >> 1)) * W + (ox >> 1)) * ldx + xcoff + cc * 8);
  *reinterpret_cast<bf16x8*>(y + p * ldy + ycoff + cc * 8) = v;
}

__global__ void upsample2x_bwd_kernel(const bf16_t* dy, int ldy, int ycoff, bf16_t* dx, int ldx, int xcoff,
                                      int accumulate, int B, int H, int W, int C, const float* dx32) {  // H,W = input dims
  const int CC = C >> 3;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)B * H * W * CC;
  if (i >= total) return;
  int cc = (int)(i % CC);
  long p = i / CC;
  int ix = (int)(p % W);
  long q = p / W;
  int iy = (int)(q % H);
  int b = (int)(q / H);
  float acc[8];
  bf16_t* d = dx + p * ldx + xcoff + cc * 8;
  if (accumulate && dx32) {                 // the partial sum of the earlier producers, kept in fp32
    const float* o32 = dx32 + p * ldx + xcoff + cc * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = o32[e];
  } else if (accumulate) {
    bf16x8 old = *reinterpret_cast<const bf16x8*>(d);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = (float)old[e];
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    long op = (long)(b * 2 * H + 2 * iy + (s >> 1)) * (2 * W) + 2 * ix + (s & 1);
    bf16x8 g = *reinterpret_cast<const bf16x8*>(dy + op * ldy + ycoff + cc * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += (float)g[e];
  }
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (bf16_t)acc[e];
  *reinterpret_cast<bf16x8*>(d) = o;
}

// ------------------------------------------------------------------ head gradient re-layout
// g[B][A][HW][P] fp32 -> dy[M = B*HW][Npad] bf16 with n = box(4A) | obj(A) | cls(nc*A); bias partials
// bpart[blk][Npad] (summed by bias_reduce).  A transpose through LDS: a block takes PREP_ROWS consecutive rows of M,
// reads each anchor's [rows][P] run of g as one contiguous stream (the rows of an image are consecutive pixels), and
// writes the rows of dy as 16-byte chunks of one contiguous run; blocks are persistent over tiles, so the bias partial
// of a block is a fixed-order sum (deterministic).
__global__ __launch_bounds__(256) void head_bwd_prep_kernel(const float* g, bf16_t* dy, float* bpart, int B, int HW,
                                                            int A, int nc, int Npad, uint32_t magic_p, int PREP_ROWS) {
  extern __shared__ float sm[];            // [PREP_ROWS][Npad + 1]
  const int P = 5 + nc;
  const int N = A * P;
  const int ldt = Npad + 1;
  const long M = (long)B * HW;
  const int ntiles = (int)((M + PREP_ROWS - 1) / PREP_ROWS);
  float bsum = 0.f;                         // thread n < Npad: running column sum over this block's tiles
  for (int i = threadIdx.x; i < PREP_ROWS * ldt; i += blockDim.x) sm[i] = 0.f;      // (padding columns stay zero)
  __syncthreads();
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const long m0 = (long)t * PREP_ROWS;
    const int b0 = (int)(m0 / HW), pix0 = (int)(m0 - (long)b0 * HW);
    // the gather is latency-bound (a block's loads of one pass depend on nothing but its indices): UNR passes' loads are
    // requested before the first value is stored (one load in flight per thread: 59 us on the 80 x 80 level)
    constexpr int UNR = 4;
    const int per_an = PREP_ROWS * P;
    for (int an = 0; an < A; ++an)
      for (int base = 0; base < per_an; base += UNR * blockDim.x) {
        float v[UNR];
        int dst[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int idx = base + u * blockDim.x + threadIdx.x;
          v[u] = 0.f; dst[u] = -1;
          if (idx < per_an) {
            const int r = (int)__umulhi((uint32_t)idx, magic_p);          // idx / P (idx < 2^13: exact)
            const int slot = idx - r * P;
            if (m0 + r < M) {
              int b = b0, pix = pix0 + r;
              while (pix >= HW) { pix -= HW; ++b; }
              v[u] = g[(((long)b * A + an) * HW + pix) * P + slot];
            }
            dst[u] = r * ldt + (slot < 4 ? 4 * an + slot : (slot == 4 ? 4 * A + an : 5 * A + an * nc + slot - 5));
          }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
          if (dst[u] >= 0) sm[dst[u]] = v[u];
      }
    __syncthreads();
    const int cpr = Npad / 8;               // 16-byte chunks per row
    for (int c = threadIdx.x; c < PREP_ROWS * cpr; c += blockDim.x) {
      const int r = c / cpr, n0 = (c - r * cpr) * 8;
      if (m0 + r < M) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)sm[r * ldt + n0 + e];
        *reinterpret_cast<bf16x8*>(dy + (m0 + r) * Npad + n0) = o;
      }
    }
    if (threadIdx.x < Npad) {
      float s = 0.f;
      for (int r = 0; r < PREP_ROWS; ++r) s += sm[r * ldt + threadIdx.x];       // rows past M hold zeros
      bsum += s;
    }
    __syncthreads();
  }
  if (threadIdx.x < Npad) bpart[(size_t)blockIdx.x * Npad + threadIdx.x] = bsum;
  (void)N;
}

// bias grads of the three heads: db_box[4A], db_obj[A], db_cls[nc*A] are consecutive in n order.
__global__ void head_bias_reduce_kernel(const float* bpart, int nblk, int Npad, float* db_box, float* db_obj,
                                        float* db_cls, int A, int nc) {
  int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);     // one wave per output channel
  int lane = threadIdx.x & 63;
  int N = A * (5 + nc);
  if (n >= N) return;
  double s = 0.0;
  for (int b = lane; b < nblk; b += 64) s += (double)bpart[(size_t)b * Npad + n];
  s = wave_sum_d(s);
  if (lane != 0) return;
  if (n < 4 * A) db_box[n] = (float)s;
  else if (n < 5 * A) db_obj[n - 4 * A] = (float)s;
  else db_cls[n - 5 * A] = (float)s;
}

// ------------------------------------------------------------------ SGD
// group id per 64-element granule: 0 bias, 1 decay, 2 norm, 255 padding
// hyper (device memory, so a captured hipGraph sees per-step schedules): lr[3] | momentum[3] | wd[3] | grad_scale | flags |
// dampening (12 floats; flags = nesterov + 2 maximize + 4 first step, small integers held in a float).  torch.optim.SGD
// (torch/optim/sgd.py _single_tensor_sgd): g = maximize ? -g : g; g += wd * p; momentum != 0: buf = first step ? g :
// mu * buf + (1 - dampening) * g; g = nesterov ? g + mu * buf : buf; p -= lr * g.  With dampening = 0 the first step needs no
// flag (buf starts at zero: mu * 0 + g = g), which is the reference's configuration (smart_sgd.yaml).
__global__ void sgd_nesterov_kernel(float* p, const float* g, float* buf, const unsigned char* gid, long n,
                                    const float* hyper) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  unsigned char grp = gid[i >> 6];
  if (grp > 2) return;
  const float lr = hyper[grp], mu = hyper[3 + grp], wd = hyper[6 + grp], gscale = hyper[9];
  const int flags = (int)hyper[10];
  const bool nesterov = (flags & 1) != 0, maximize = (flags & 2) != 0, first = (flags & 4) != 0;
  const float undamped = 1.0f - hyper[11];
  f32x4 pv = *reinterpret_cast<const f32x4*>(p + i);
  f32x4 gv = kod_load_once<f32x4>(g + i);
  f32x4 bv = *reinterpret_cast<const f32x4*>(buf + i);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float gg = gv[e] * gscale;
    if (maximize) gg = -gg;
    if (wd != 0.f) gg = gg + wd * pv[e];
    float b = (first || mu == 0.f) ? gg : mu * bv[e] + undamped * gg;      // (undamped = 1: mu * buf + g, bit for bit)
    bv[e] = b;
    pv[e] = pv[e] - lr * (nesterov ? gg + mu * b : b);
  }
  *reinterpret_cast<f32x4*>(p + i) = pv;
  *reinterpret_cast<f32x4*>(buf + i) = bv;
}

__global__ void fill_u32_kernel(uint32_t* p, uint32_t v, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

__global__ void stamp_kernel(unsigned long long* dst) { *dst = wall_clock64(); }

// small host -> device upload as a KERNEL reading pinned host memory through its device mapping (16 bytes per lane):
// on this stack an async copy queued on a compute stream costs that stream ~0.15 ms whatever its size (engine hand-over),
// a kernel that pulls the same bytes over PCIe a few microseconds
__global__ __launch_bounds__(256) void pull_host_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long n16) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
}

}  // namespace

extern "C" {

int kodhip_nchw_to_nhwc4(const float* x, void* y, int B, int C, int H, int W, hipStream_t stream) {
  KOD_CHECK_ARG(x && y && B > 0 && C > 0 && C <= 4 && H > 0 && W > 0, "nchw_to_nhwc4: bad args");
  long n = (long)B * H * W;
  if (((long)H * W) % 4 == 0 && ((uintptr_t)x % 16) == 0)
    hipLaunchKernelGGL(nchw_to_nhwc4_x4_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, stream, x, (bf16_t*)y, B, C, (long)H * W);
  else
    hipLaunchKernelGGL(nchw_to_nhwc4_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, x, (bf16_t*)y, B, C, (long)H * W);
  KOD_LAUNCH_CHECK("nchw_to_nhwc4");
  return KOD_OK;
}

// descs: device array of nlayers PackDesc (13 x int64 each); total_blocks = sum over descriptors of
// ceil(N*Cin*KH*KW / 256).
int kodhip_pack_weights(const float* master, void* fpack, void* dpack, const void* descs, int nlayers,
                        long total_blocks, hipStream_t stream) {
  KOD_CHECK_ARG(master && fpack && descs && nlayers > 0 && total_blocks > 0, "pack_weights: bad args");
  hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)total_blocks), dim3(256), 0, stream, master, (bf16_t*)fpack,
                     (bf16_t*)dpack, (const PackDesc*)descs, nlayers);
  KOD_LAUNCH_CHECK("pack_weights");
  return KOD_OK;
}
int kodhip_pack_desc_bytes(void) { return (int)sizeof(PackDesc); }

int kodhip_maxpool5_fwd(const void* x, int ldx, int xcoff, void* y, int ldy, int ycoff, void* idx,
                        int B, int H, int W, int C, hipStream_t stream) {
  KOD_CHECK_ARG(x && y && idx && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && xcoff % 8 == 0 && ycoff % 8 == 0,
                "maxpool5_fwd: bad args");
  long n = (long)B * ((H + 3) / 4) * ((W + 3) / 4) * (C / 4);      // a 4 x 4 block of outputs x 4 channels per thread
  hipLaunchKernelGGL(maxpool5_fwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, (const bf16_t*)x, ldx, xcoff,
                     (bf16_t*)y, ldy, ycoff, (unsigned char*)idx, B, H, W, C);
  KOD_LAUNCH_CHECK("maxpool5_fwd");
  return KOD_OK;
}

// dx_f32 (may be NULL): fp32 shadow of dx (same indexing) holding the partial sum the earlier producers left - read
// instead of dx's bf16 content, so the total is rounded once (see kodhip_conv_dgrad's `accumulate`)
int kodhip_maxpool5_bwd(const void* dy, int ldy, int ycoff, const void* idx, void* dx, int ldx, int xcoff,
                        int B, int H, int W, int C, const float* dx_f32, hipStream_t stream) {
  KOD_CHECK_ARG(dy && dx && idx && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && xcoff % 8 == 0 && ycoff % 8 == 0,
                "maxpool5_bwd: bad args");
  long n = (long)B * ((H + 3) / 4) * ((W + 3) / 4) * (C / 4);      // a 4 x 4 block of inputs x 4 channels per thread
  hipLaunchKernelGGL(maxpool5_bwd_kernel, dim3(cdiv(n, POOL_BWD_THREADS)), dim3(POOL_BWD_THREADS), 0, stream, (const bf16_t*)dy, ldy, ycoff,
                     (const unsigned char*)idx, (bf16_t*)dx, ldx, xcoff, B, H, W, C, dx_f32);
  KOD_LAUNCH_CHECK("maxpool5_bwd");
  return KOD_OK;
}

// Max-pool K x K / stride 1 / pad K / 2 for any odd K <= 15 (SPPFBottleneck's kernel_sizes, kod/nn/layers/sppf.py:27-67):
// K = 5 takes the tuned kernels (kodhip_maxpool5_*), every other window the plain ones.  Same idx / dx_f32 conventions.
int kodhip_maxpool_fwd(const void* x, int ldx, int xcoff, void* y, int ldy, int ycoff, void* idx,
                       int B, int H, int W, int C, int K, hipStream_t stream) {
  if (K == 5) return kodhip_maxpool5_fwd(x, ldx, xcoff, y, ldy, ycoff, idx, B, H, W, C, stream);
  KOD_CHECK_ARG(x && y && idx && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && xcoff % 8 == 0 && ycoff % 8 == 0, "maxpool_fwd: bad args");
  KOD_CHECK_ARG(K >= 1 && K <= 15 && (K & 1), "maxpool_fwd: window %d (odd sizes up to 15)", K);
  const long n = (long)B * H * W * (C / 8);
  hipLaunchKernelGGL(maxpool_k_fwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, (const bf16_t*)x, ldx, xcoff, (bf16_t*)y, ldy, ycoff,
                     (unsigned char*)idx, B, H, W, C, K);
  KOD_LAUNCH_CHECK("maxpool_fwd");
  return KOD_OK;
}

int kodhip_maxpool_bwd(const void* dy, int ldy, int ycoff, const void* idx, void* dx, int ldx, int xcoff,
                       int B, int H, int W, int C, int K, const float* dx_f32, hipStream_t stream) {
  if (K == 5) return kodhip_maxpool5_bwd(dy, ldy, ycoff, idx, dx, ldx, xcoff, B, H, W, C, dx_f32, stream);
  KOD_CHECK_ARG(dy && dx && idx && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && xcoff % 8 == 0 && ycoff % 8 == 0, "maxpool_bwd: bad args");
  KOD_CHECK_ARG(K >= 1 && K <= 15 && (K & 1), "maxpool_bwd: window %d (odd sizes up to 15)", K);
  const long n = (long)B * H * W * (C / 8);
  hipLaunchKernelGGL(maxpool_k_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, (const bf16_t*)dy, ldy, ycoff, (const unsigned char*)idx,
                     (bf16_t*)dx, ldx, xcoff, B, H, W, C, K, dx_f32);
  KOD_LAUNCH_CHECK("maxpool_bwd");
  return KOD_OK;
}

int kodhip_upsample2x_fwd(const void* x, int ldx, int xcoff, void* y, int ldy, int ycoff,
                          int B, int H, int W, int C, hipStream_t stream) {
  KOD_CHECK_ARG(x && y && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && xcoff % 8 == 0 && ycoff % 8 == 0,
                "upsample2x_fwd: bad args");
  long n = (long)B * 4 * H * W * (C / 8);
  hipLaunchKernelGGL(upsample2x_fwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, (const bf16_t*)x, ldx, xcoff,
                     (bf16_t*)y, ldy, ycoff, B, H, W, C);
  KOD_LAUNCH_CHECK("upsample2x_fwd");
  return KOD_OK;
}

int kodhip_upsample2x_bwd(const void* dy, int ldy, int ycoff, void* dx, int ldx, int xcoff, int accumulate,
                          int B, int H, int W, int C, const float* dx_f32, hipStream_t stream) {
  KOD_CHECK_ARG(dy && dx && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && xcoff % 8 == 0 && ycoff % 8 == 0,
                "upsample2x_bwd: bad args");
  long n = (long)B * H * W * (C / 8);
  hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, (const bf16_t*)dy, ldy, ycoff,
                     (bf16_t*)dx, ldx, xcoff, accumulate, B, H, W, C, dx_f32);
  KOD_LAUNCH_CHECK("upsample2x_bwd");
  return KOD_OK;
}

// workspace: bias partials, 2048 * Npad floats
int kodhip_head_bwd_prep(const float* g, void* dy, float* workspace, float* db_box, float* db_obj, float* db_cls,
                         int B, int HW, int A, int nc, int Npad, hipStream_t stream) {
  KOD_CHECK_ARG(g && dy && workspace && db_box && db_obj && db_cls, "head_bwd_prep: null pointer");
  KOD_CHECK_ARG(Npad % 8 == 0 && Npad >= A * (5 + nc) && Npad <= 256, "head_bwd_prep: bad Npad");
  KOD_CHECK_ARG(nc > 0 && 5 + nc <= 128, "head_bwd_prep: bad class count");
  long M = (long)B * HW;
  const int PREP_ROWS = Npad <= 128 ? 64 : 32;          // rows per tile: [rows][Npad + 1] floats of LDS (<= 33 KB)
  int grid = (int)((M + PREP_ROWS - 1) / PREP_ROWS);
  if (grid > 2048) grid = 2048;           // 8 blocks per CU: the load / store phases of a tile are latency-bound
  hipLaunchKernelGGL(head_bwd_prep_kernel, dim3(grid), dim3(256), PREP_ROWS * (Npad + 1) * sizeof(float), stream, g,
                     (bf16_t*)dy, workspace, B, HW, A, nc, Npad, magic_u32((uint32_t)(5 + nc)), PREP_ROWS);
  KOD_LAUNCH_CHECK("head_bwd_prep");
  hipLaunchKernelGGL(head_bias_reduce_kernel, dim3(cdiv(A * (5 + nc), 4)), dim3(256), 0, stream,
                     (const float*)workspace, grid, Npad, db_box, db_obj, db_cls, A, nc);
  KOD_LAUNCH_CHECK("head_bias_reduce");
  return KOD_OK;
}

// hyper: DEVICE pointer to 10 floats = lr[3], momentum[3], weight_decay[3], grad_scale  (groups: bias, decay, norm)
int kodhip_sgd_nesterov(float* params, const float* grads, float* momentum_buf, const void* group_ids,
                        long n, const float* hyper, hipStream_t stream) {
  KOD_CHECK_ARG(params && grads && momentum_buf && group_ids && hyper && n > 0 && n % 64 == 0, "sgd_nesterov: bad args");
  hipLaunchKernelGGL(sgd_nesterov_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, stream, params, grads, momentum_buf,
                     (const unsigned char*)group_ids, n, hyper);
  KOD_LAUNCH_CHECK("sgd_nesterov");
  return KOD_OK;
}

// debug: the device's constant-rate clock (100 MHz) written by a one-lane kernel - a time stamp INSIDE a captured,
// replayed step, where HIP events cannot be read back and a profiler changes the schedule (bench.py stamp_report)
int kodhip_debug_stamp(unsigned long long* dst, hipStream_t stream) {
  KOD_CHECK_ARG(dst, "debug_stamp: null");
  hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, stream, dst);
  KOD_LAUNCH_CHECK("debug_stamp");
  return KOD_OK;
}

// dst (device) <- src (PINNED host memory, e.g. a torch pin_memory() tensor), bytes a multiple of 16, both 16-byte aligned:
// the per-step tables of the data path (compositing descriptors, the batch's target block) without a copy-engine hand-over
// on the step's stream.  Falls back to hipMemcpyAsync when the host memory has no device mapping.
int kodhip_pull_from_host(void* dst, const void* src_pinned, long bytes, hipStream_t stream) {
  KOD_CHECK_ARG(dst && src_pinned && bytes > 0 && bytes % 16 == 0 && ((uintptr_t)dst % 16) == 0 && ((uintptr_t)src_pinned % 16) == 0,
                "pull_from_host: bad args (16-byte granularity)");
  void* mapped = nullptr;
  if (hipHostGetDevicePointer(&mapped, const_cast<void*>(src_pinned), 0) != hipSuccess || !mapped) {
    (void)hipGetLastError();
    hipError_t e = hipMemcpyAsync(dst, src_pinned, (size_t)bytes, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) { kodhip_set_error("pull_from_host: %s", hipGetErrorString(e)); return (int)e; }
    return KOD_OK;
  }
  const long n16 = bytes / 16;
  const int grid = (int)(cdiv(n16, 256) < 256 ? cdiv(n16, 256) : 256);
  hipLaunchKernelGGL(pull_host_kernel, dim3(grid), dim3(256), 0, stream, (const uint4*)mapped, (uint4*)dst, n16);
  KOD_LAUNCH_CHECK("pull_from_host");
  return KOD_OK;
}

int kodhip_fill_u32(void* p, uint32_t value, long n, hipStream_t stream) {
  KOD_CHECK_ARG(p && n > 0, "fill_u32: bad args");
  hipLaunchKernelGGL(fill_u32_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, (uint32_t*)p, value, n);
  KOD_LAUNCH_CHECK("fill_u32");
  return KOD_OK;
}

}  // extern "C"
