// Implicit-GEMM convolution for gfx950 (CDNA4): bf16 MFMA 32x32x16, fp32 accumulate.
//
// One kernel serves
//   * forward conv  (kod.nn conv+BN+SiLU units: torchvision Conv2dNormActivation as used at
//     kod/nn/layers/csp.py:30-46,81-89, kod/nn/backbones/yolov5.py:44-52,102-110, ...)  -> mode RAW:
//     bf16 pre-BN output + per-channel sum / sum-of-squares partials for train-mode BatchNorm
//   * the three biased 1x1 head convs per level (kod/nn/heads/yolov5.py:12-136) fused into one GEMM
//     -> mode HEAD: fp32 output scattered into [B, A, h, w, 5+nc]
//   * data gradient (aten::convolution_backward dX) as a gather-form transposed conv -> mode PLAIN
//
// Data layout: activations channels-last [B, H, W, ld] bf16 (a tensor may be a channel slice of a
// wider concat buffer: ld / channel offset), weights pre-packed [N][Kp] with k = (kh, kw, ci),
// k-contiguous, so both MFMA operands are 16-byte k-runs.
//
// Tiling: tile {128,256}(M pixels) x {32,64,128}(out channels) x 32(K), 4 or 8 waves.  FAST path (every layer whose
// per-tap channel count is a multiple of 32 after padding, i.e. all of yv5n/s/m): operands go HBM -> LDS by LDS-DMA
// (buffer_load ... lds) through a 2- or 3-stage ring, rows unpadded with a source-side XOR swizzle, counted vmcnt +
// raw s_barrier; generic path (KODHIP_NO_FAST, operands beyond a 32-bit buffer range): global -> registers -> LDS,
// double buffered, 80-byte padded rows.  The MFMA is issued "transposed" (A operand = weights, B operand = pixels)
// so each lane ends up with 4 consecutive output channels per accumulator quad and the epilogue can stage the tile
// through LDS with ds_write_b64 and leave the chip as full 16-byte channel-contiguous stores.
// Blocks are persistent over M tiles (fixed N tile) so BatchNorm partial statistics are reduced in
// registers and written once per block; block ids are laid out so the N tiles that share an M tile
// land on the same XCD (same L2).
#include "kodhip_common.h"
#include <stdlib.h>

namespace {

enum { MODE_RAW = 0, MODE_PLAIN = 1, MODE_HEAD = 2, MODE_PLAIN_BN = 3 };

// MODE_PLAIN_BN: a data-gradient launch that is the LAST writer of some conv units' output gradients also produces
// their BatchNorm-backward reduction (sum dz, sum dz*y per channel) in its epilogue - the tile it has just written
// is the dA operand of that reduction, so the separate pass over (dA, y) disappears.  A segment = the channel range
// of the launch's output that belongs to one such unit.
constexpr int MAX_SEG = 3;

struct ConvArgs {
  const bf16_t* x;
  const bf16_t* w;
  bf16_t* y;
  float* stats;
  const float* bias;
  float* head_out;
  int B, Hs, Ws, ldx, xcoff, Cin;
  int cin_step;                // channels per tap on the packed K axis = round_up(Cin, 32): k = tap * cin_step + ci, weights of
                               // ci >= Cin are zero (Cin = 48: the last K step of a tap reads 16 channels past the slice - finite
                               // activations of the neighbouring pixel / slice, or zeros past the buffer - times zero weights)
  int Ho, Wo, M;
  int N, K, Kp;
  int KH, KW;
  int mul_h, mul_w, add_h, add_w, tap_sign, sh_shift, sw_shift;
  int ldy, ycoff, accumulate;
  // PLAIN: fp32 shadow of the output buffer (same pixel / channel indexing, row stride ldy floats) for activation
  // gradients with several producers.  f32_mode: 0 off; 1 = also write this launch's fp32 values there (first
  // producer); 2 / 3 = a later producer: the shadow's partial sum is added to the MFMA accumulators BEFORE the single bf16
  // rounding and the bf16 output is overwritten with the rounded running sum (2 also stores the sum back to the shadow,
  // 3 = last producer, does not); 4 = the partial is the bf16 output itself (left there by exact producers - copies such
  // as a residual pass-through): added in fp32 before the rounding, no shadow
  float* y32;
  int f32_mode;
  int out_mul, out_off_y, out_off_x, out_H, out_W;   // PLAIN: output pixel (oy,ox) -> (oy*mul+off_y, ox*mul+off_x) of an out_H x out_W image
  int d2s_C;                   // PLAIN, folded stride-2 data gradient: output column n = class * d2s_C + channel, class (py,px)
                               // = (n / d2s_C) >> 1, & 1 is the pixel's parity offset (depth-to-space epilogue); 0 = off
  int d2s_skip;                // folded form: column tiles that lie inside the py = 0 classes (n < 2 * d2s_C) stop after the
                               // dy = 0 taps - the first half of the K steps; their other weights are zero by construction
  int head_A, head_P, head_nc;
  uint32_t magic_hp;           // HEAD: magic of head_P
  uint32_t magic_cin, magic_kw;
  int tiles_m, tiles_n, groups_m, stats_slots;
  float rcp_hwo, rcp_wo;       // reciprocals for the row -> (b, oy, ox) decomposition (m < 2^24: one fix-up step)
  uint32_t x_bytes, w_bytes;   // FAST path: byte extents of the gather source / weight pack (buffer descriptors)
  // dual-source pointwise gather (FAST, 1x1): K = [channels of x through w | channels of x2 through w2], both sources
  // with the geometry of x (same ld / channel offset / channel count / Kp).  The data gradients of two convs that read
  // the same tensor (a CSP layer's main and short convs) become ONE launch that writes dX once: no second launch, no
  // read-modify-write accumulation pass over dX.  nk1 = K steps taken from the first source (0 = single source).
  const bf16_t* x2;
  const bf16_t* w2;
  int nk1;
  int wide_px;                 // FAST path: a tap reads Cin = wide_px * ldx channels = wide_px consecutive pixels (stem)
  int pointwise;               // FAST path: 1x1 / stride 1 / no padding - gather row m is input pixel m (set by the launcher)
  // MODE_PLAIN_BN (stats_slots = slot capacity of every seg_part buffer)
  int nseg, slot_base, slot_used;
  int seg_begin[MAX_SEG], seg_end[MAX_SEG], seg_ldr[MAX_SEG], seg_C[MAX_SEG];
  const bf16_t* seg_raw[MAX_SEG];      // the unit's pre-BN output y [M_out][ldr]
  const float* seg_aff[MAX_SEG];       // scale[C] | shift[C]
  float* seg_part[MAX_SEG];            // [2][C][stats_slots]: sum dz | sum dz*y
};

// q = n / d, r = n % d via a float reciprocal + fix-up (exact: the loops absorb the fp32 rounding of large n)
// instead of a 40-instruction integer divide
__device__ __forceinline__ void fast_divmod(int n, int d, float rcp, int& q, int& r) {
  q = (int)((float)n * rcp);
  r = n - q * d;
  while (r < 0) { --q; r += d; }
  while (r >= d) { ++q; r -= d; }
}

constexpr int BK = 32;

template <int N>
__device__ __forceinline__ void wait_vm_imm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// launch bounds: 4 blocks per CU (one wave of each on every SIMD) => at most 128 VGPRs, so that one block's LDS /
// global phases overlap another block's MFMA phase (measured: the phases of a single block do not overlap)
// ROW3 (FAST, 3x3 / stride 1 / pad 1, 128-pixel tiles): a K step is (kernel row kh, 32-channel chunk) and covers the
// row's THREE taps - the source row segment [m0 - 1, m0 + BM] is staged once (BM + 16 rows with DMA granularity) and
// the taps kw = 0, 1, 2 read it at row offsets 0, 1, 2 (their weight tiles are staged side by side); what the
// zero-filling buffer load did per tap for the left / right image border becomes a per-lane mask on the pixel
// fragments of taps 0 and 2.  Per unit of MFMA work the L2 -> LDS fill drops by 30-50 % (these layers are bound by it)
// and there is one barrier per three taps.
// STEM (FAST, the 6x6 / stride 2 / pad 2 stem in its wide-pixel form: 8-channel pixel PAIRS, a K step = one kernel row =
// four consecutive pairs, the fourth with zero weights): the generic form stages 4 pairs (64 B) per output pixel and
// kernel row although neighbouring output pixels share three of them; here a K step stages ONE run of BM + 3 pairs (16-byte
// rows; output pixel i, tap j reads staged row i + j) - a quarter of the L2 -> LDS fill of a layer that is bound by it.
// Rows are taken in flattened (b, oy, ox) order with each staged row addressed by its own pixel, so tiles may cross image
// rows; the taps that would then read the neighbouring image row's pairs (j = 0 at ox = 0, j = 2 at ox = Wo - 1) are
// zero-padding and masked on the fragments, as in ROW3.
template <int BM, int BN, int WAVES_M, int WAVES_N, int MODE, bool FAST, bool ROW3 = false, bool F32ACC = false, bool STEM = false>
__device__ __forceinline__ void conv_igemm_body(const ConvArgs& a, const int bid) {
  constexpr int NT = 64 * WAVES_M * WAVES_N;
  constexpr int NW = NT / 64;
  static_assert(FAST || (BM == 128 && NT == 256), "register-staged path is written for 128-row tiles / 4 waves");
  // staged row: generic path pads to 80 bytes (conflict-free b128 reads); the FAST path stages by LDS-DMA, whose
  // image must be lane-linear (no padding) - conflicts are removed by an XOR swizzle applied on the SOURCE side
  constexpr int LDS_ROW = FAST ? BK : BK + 8;
  constexpr int WM = BM / WAVES_M;          // pixels per wave
  constexpr int WN = BN / WAVES_N;          // channels per wave
  constexpr int TM = WM / 32;
  constexpr int TN = WN / 32;
  constexpr int B_CHUNKS = BN * 4;          // 16-byte chunks in the weight tile
  constexpr int B_PER_THREAD = (B_CHUNKS + NT - 1) / NT;
  static_assert(!ROW3 || (FAST && BM == 128), "ROW3 is a FAST-path form for 128-pixel tiles");
  constexpr int A3_ROWS = BM + 16;                   // ROW3: staged source rows (BM + 2 used)
  static_assert(!STEM || (FAST && BM == 128 && !ROW3 && MODE == MODE_RAW), "STEM is a forward FAST-path form for 128-pixel tiles");
  constexpr int AS_ROWS = BM + 64;                   // STEM: staged 16-byte pair rows (BM + 3 used): 3 DMA instructions
  constexpr int STAGE_ELEMS = STEM ? AS_ROWS * 8 + BN * LDS_ROW : (ROW3 ? (A3_ROWS + 3 * BN) * LDS_ROW : (BM + BN) * LDS_ROW);
  constexpr int CS_ROW = BN + 8;            // epilogue staging row (bf16)
  // LDS stages.  The fill of the wide tiles is latency-bound - a CU moves (bytes in flight) / (L2-or-HBM latency) -
  // so 256-row tiles (2 blocks per CU) run a 3-stage ring; 128-row tiles keep 2 stages and 4 blocks per CU (a third
  // stage costs 128x128 its 4th block, and measured nothing on the narrow tiles, whose 3x3 layers are bound by the
  // 9x im2col re-read through L2 instead).
  constexpr int NST = (FAST && BM == 256 && !ROW3) ? 3 : 2;
  constexpr int LDS_ELEMS = (NST * STAGE_ELEMS > BM * CS_ROW) ? NST * STAGE_ELEMS : BM * CS_ROW;
  __shared__ __attribute__((aligned(16))) bf16_t lds[LDS_ELEMS];
  float* sred = reinterpret_cast<float*>(lds);     // BN partial statistics reuse the staging area after the tile loop

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;

  // XCD-aware block -> (m group, n tile): blocks b and b+8 share an XCD.
  const int xcd = bid & 7;
  const int j = bid >> 3;
  const int nt = j % a.tiles_n;
  const int gm = (j / a.tiles_n) * 8 + xcd;
  if (gm >= a.groups_m) return;
  const int n0 = nt * BN;

  // staging assignment
  const int a_chunk = tid & 3;              // which 8-element k chunk of the 32-wide K tile
  const int a_row = tid >> 2;               // rows a_row and a_row + 64
  const int HWs = a.Hs * a.Ws;
  const int HWo = a.Ho * a.Wo;

  float ssum[8], ssq[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) ssum[i] = ssq[i] = 0.f;
  float bn_run = 0.f;                        // MODE_PLAIN_BN: running sum of (channel tid>>1, statistic tid&1)

  int nk_steps = STEM ? a.KH : (ROW3 ? 3 * (a.cin_step / BK) : (a.nk1 ? 2 * a.nk1 : a.Kp / BK));
  if constexpr ((MODE == MODE_PLAIN || MODE == MODE_PLAIN_BN) && FAST && !ROW3 && !STEM) {
    // folded stride-2 data gradient (prep_dgrad_s2f): classes (0, px) meet only the taps dy = 0, K steps [0, nk / 2)
    if (a.d2s_skip && n0 + BN <= 2 * a.d2s_C) nk_steps >>= 1;
  }
  const int nk = nk_steps;
  // FAST path (Cin % 32 == 0, unit tap stride, no K tail): operands come through raw buffer loads - the tap of a
  // K step is wave-uniform (scalar registers), invalid (padding) elements are fetched from an out-of-range offset
  // that the buffer unit returns as zeros, so a step costs ~10 VALU instead of ~110 and has no branches.
  __amdgpu_buffer_rsrc_t rs_x, rs_w, rs_x2, rs_w2;
  if constexpr (FAST) {
    rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
    rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);
    rs_x2 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.nk1 ? a.x2 : a.x), 0, a.x_bytes, 0x00020000);
    rs_w2 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.nk1 ? a.w2 : a.w), 0, a.w_bytes, 0x00020000);
  }

  // m tiles -> groups, in BANDS per XCD: XCD k owns a contiguous range of tiles (sized in proportion to its share of the
  // groups: with one tile per group nobody gets two) and its groups sweep the band together (group q of the XCD takes tiles
  // begin + q, + gk, ...).  The tiles an XCD works on at any time are then neighbours in the image, and the input rows that
  // vertically adjacent tiles share (a 3x3 layer reads every input row for three output rows, the stem for three of its
  // own) come out of that XCD's L2.  With the round-robin order this replaces (tile mt on XCD mt % 8) FETCH_SIZE showed no
  // vertical reuse at all: 3x3 stride-1 layers fetched 3.1 x their input, stride-2 layers 1.5 - 2.1 x, the stem 3.4 x
  // (1.9 GB of the forward pass's 6.2 GB of reads per step).  The Infinity Cache served those re-reads, so the step time
  // is the same; what changes is the traffic over the fabric.
  const int gq = gm >> 3, gk = (a.groups_m - xcd + 7) >> 3;            // this group's index among its XCD's gk groups
  int gbefore = 0;                                                     // groups of the XCDs before this one
  for (int i = 0; i < xcd; ++i) gbefore += (a.groups_m - i + 7) >> 3;
  const int band_begin = (int)((long)a.tiles_m * gbefore / a.groups_m);
  const int band_end = (int)((long)a.tiles_m * (gbefore + gk) / a.groups_m);
  for (int mt = band_begin + gq; mt < band_end; mt += gk) {
    const int m0 = mt * BM;
    // ---- per-thread gather rows: base pointer + tap-validity bit mask, computed once per tile so the
    //      K loop only adds a per-step tap offset (keeps the VALU out of the MFMA's way)
    int rbase[2], rby[2], rbx[2];
    bool rvalid[2];
    const bf16_t* rptr[2];
    uint32_t rmask[2];
    const bool linear_taps = (a.sh_shift | a.sw_shift) == 0;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      int m = m0 + a_row + r * 64;
      rvalid[r] = m < a.M;
      int mm = rvalid[r] ? m : 0;
      int b, rem, oy, ox;
      fast_divmod(mm, HWo, a.rcp_hwo, b, rem);
      fast_divmod(rem, a.Wo, a.rcp_wo, oy, ox);
      rbase[r] = b * HWs;
      rby[r] = oy * a.mul_h + a.add_h;
      rbx[r] = ox * a.mul_w + a.add_w;
      rptr[r] = a.x + ((long)(rbase[r] + rby[r] * a.Ws + rbx[r]) * a.ldx + a.xcoff);
      uint32_t xm = 0, mk = 0;
      for (int kw = 0; kw < a.KW; ++kw)
        if ((unsigned)(rbx[r] + a.tap_sign * kw) < (unsigned)a.Ws) xm |= 1u << kw;
      for (int kh = 0; kh < a.KH; ++kh)
        if ((unsigned)(rby[r] + a.tap_sign * kh) < (unsigned)a.Hs) mk |= xm << (kh * a.KW);
      rmask[r] = rvalid[r] ? mk : 0u;
    }
    const bf16_t* bptr[B_PER_THREAD];
#pragma unroll
    for (int q = 0; q < B_PER_THREAD; ++q) {
      int c = tid + q * NT;
      int n = n0 + (c >> 2);
      bptr[q] = (c < B_CHUNKS && n < a.N) ? a.w + (size_t)n * a.Kp + (c & 3) * 8 : nullptr;
    }

    // FAST path: LDS-DMA staging.  One buffer_load..lds moves 64 lanes x 16 B = 16 rows x 64 B into a linear LDS
    // block; wave w owns rows [32w, 32w+32) of the pixel tile (2 instructions) and its share of the weight tile.
    // LDS slot (row, c') holds global chunk c = c' ^ ((row >> 2) & 3)  (the fragment reads undo the same XOR).
    constexpr int B_INSTR = BN / 16;                  // weight-tile DMA instructions per K step (per block)
    constexpr int B_PER_WAVE = (B_INSTR + NW - 1) / NW;
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    constexpr int A_PER_WAVE = BM / (16 * NW);         // pixel-tile DMA instructions per wave (16 rows each)
    static_assert(!FAST || A_PER_WAVE * 16 * NW == BM, "pixel tile rows must divide over the waves' DMA instructions");
    uint32_t dvoff[A_PER_WAVE > 0 ? A_PER_WAVE : 1], dmask[A_PER_WAVE > 0 ? A_PER_WAVE : 1], dbvoff[B_PER_WAVE];
    int f_tap = 0, f_kh = 0, f_kw = 0, f_ci = 0;      // wave-uniform K-step state
    if constexpr (FAST) {
#pragma unroll
      for (int i = 0; i < ((ROW3 || STEM) ? 0 : A_PER_WAVE); ++i) {
        int row = (uwave * A_PER_WAVE + i) * 16 + (lane >> 2);
        int m = m0 + row;
        bool valid = m < a.M;
        int mm = valid ? m : 0;
        int chunk = (lane & 3) ^ ((row >> 2) & 3);
        if (a.pointwise) {
          // 1x1 / stride 1 / no padding (39 of the 57 convs and their data gradients): output pixel m reads input pixel m -
          // no (b, oy, ox) decomposition, no tap masks (the two float-reciprocal divisions and the mask loops below are
          // most of a tile's set-up, which a one- or two-K-step tile of the shallow layers pays as often as it computes)
          dvoff[i] = (uint32_t)(((long)mm * a.ldx + a.xcoff + chunk * 8) * 2);
          dmask[i] = valid ? 1u : 0u;
          continue;
        }
        int b, rem, oy, ox;
        fast_divmod(mm, HWo, a.rcp_hwo, b, rem);
        fast_divmod(rem, a.Wo, a.rcp_wo, oy, ox);
        int by = oy * a.mul_h + a.add_h, bx = ox * a.mul_w + a.add_w;
        dvoff[i] = (uint32_t)(((long)(b * HWs + by * a.Ws + bx) * a.ldx + a.xcoff + chunk * 8) * 2);
        // wide pixels (stem: a 32-value K step spans 4 consecutive 8-channel pixel pairs): this lane's 16-byte
        // chunk belongs to pixel bx + chunk*8/ldx, which has its own left/right padding test
        const int bxl = a.wide_px > 1 ? bx + (chunk * 8) / a.ldx : bx;
        uint32_t xm = 0, mk = 0;
        for (int kw = 0; kw < a.KW; ++kw)
          if ((unsigned)(bxl + a.tap_sign * kw) < (unsigned)a.Ws) xm |= 1u << kw;
        for (int kh = 0; kh < a.KH; ++kh)
          if ((unsigned)(by + a.tap_sign * kh) < (unsigned)a.Hs) mk |= xm << (kh * a.KW);
        dmask[i] = valid ? mk : 0u;
      }
#pragma unroll
      for (int q = 0; q < B_PER_WAVE; ++q) {
        int row = (uwave * B_PER_WAVE + q) * 16 + (lane >> 2);
        int n = n0 + row;
        int chunk = (lane & 3) ^ ((row >> 2) & 3);
        dbvoff[q] = (row < BN && n < a.N) ? (uint32_t)(((size_t)n * a.Kp + chunk * 8) * 2) : 0xFFFFFFF0u;
      }
    }
    auto dma_tile = [&](int kt, int buf) {
      const uint32_t soff = (uint32_t)((a.tap_sign * (f_kh * a.Ws + f_kw) * a.ldx + f_ci) * 2);
      char* As = reinterpret_cast<char*>(lds + buf * STAGE_ELEMS);
      char* Bs = As + BM * LDS_ROW * 2;
      // dual-source pointwise gather: K steps nk1 .. 2*nk1-1 come from the second source / weight pack
      const bool second = a.nk1 != 0 && kt >= a.nk1;
      const int kw_step = second ? kt - a.nk1 : kt;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(KOD_ABL_NODMA)   // (the host pass only needs the kernel's stub)
      const __amdgpu_buffer_rsrc_t rx = second ? rs_x2 : rs_x, rw = second ? rs_w2 : rs_w;
#pragma unroll
      for (int i = 0; i < A_PER_WAVE; ++i) {
        uint32_t vo = ((dmask[i] >> f_tap) & 1u) ? dvoff[i] + soff : 0xFFFFFFF0u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(As + (uwave * A_PER_WAVE + i) * 16 * 64),
                                                 16, vo, 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < B_PER_WAVE; ++q) {
        int blk = uwave * B_PER_WAVE + q;
        if (blk < B_INSTR)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(Bs + blk * 16 * 64),
                                                   16, dbvoff[q], kw_step * (BK * 2), 0, 0);
      }
#else
      (void)soff; (void)As; (void)Bs; (void)kt;
#endif
      f_ci += BK;
      if (f_ci >= a.cin_step) {
        f_ci = 0; ++f_tap;
        if (++f_kw == a.KW) { f_kw = 0; ++f_kh; }
        if (a.nk1 != 0) f_tap = f_kh = f_kw = 0;      // dual source: one tap, the channel counter restarts on the second source
      }
    };

    // ---- ROW3 staging state (see the header comment of this function)
    constexpr int A3_INSTR = A3_ROWS / 16;
    constexpr int A3_PW = (A3_INSTR + NW - 1) / NW;
    uint32_t a3_voff[ROW3 ? A3_PW : 1], a3_mask[ROW3 ? A3_PW : 1];
    uint32_t xmask = 0;                               // bit 2*jj: tap kw=0 valid, bit 2*jj+1: tap kw=2 valid (this lane's pixel)
    if constexpr (ROW3) {
#pragma unroll
      for (int i = 0; i < A3_PW; ++i) {
        const int blk = uwave + i * NW;               // DMA instruction = 16 staged rows
        const int r = blk * 16 + (lane >> 2);
        const long q = (long)m0 - 1 + r;              // source pixel of the centre tap (flattened b, y, x)
        const bool valid = blk < A3_INSTR && r < BM + 2 && q >= 0 && q < a.M;
        const int qq = valid ? (int)q : 0;
        int b, rem, oy, ox;
        fast_divmod(qq, HWo, a.rcp_hwo, b, rem);
        fast_divmod(rem, a.Wo, a.rcp_wo, oy, ox);
        const int chunk = (lane & 3) ^ ((r >> 2) & 3);
        a3_voff[i] = (uint32_t)(((long)qq * a.ldx + a.xcoff + chunk * 8) * 2);
        uint32_t mk = 0;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
          if ((unsigned)(oy + a.tap_sign * (kh - 1)) < (unsigned)a.Hs) mk |= 1u << kh;
        a3_mask[i] = valid ? mk : 0u;
      }
#pragma unroll
      for (int jj = 0; jj < TM; ++jj) {
        const int p = m0 + wm * WM + jj * 32 + (lane & 31);
        int b, rem, oy, ox;
        fast_divmod(p < a.M ? p : 0, HWo, a.rcp_hwo, b, rem);
        fast_divmod(rem, a.Wo, a.rcp_wo, oy, ox);
        // tap kw reads source column ox + tap_sign * (kw - 1)
        const bool ok0 = a.tap_sign > 0 ? ox != 0 : ox != a.Wo - 1;
        const bool ok2 = a.tap_sign > 0 ? ox != a.Wo - 1 : ox != 0;
        xmask |= (ok0 ? 1u : 0u) << (2 * jj) | (ok2 ? 1u : 0u) << (2 * jj + 1);
      }
    }
    auto dma_tile3 = [&](int buf) {
      // step state: f_kh = kernel row, f_ci = first channel of the chunk
      const uint32_t soff = (uint32_t)((a.tap_sign * (f_kh - 1) * a.Ws * a.ldx + f_ci) * 2);
      char* As = reinterpret_cast<char*>(lds + buf * STAGE_ELEMS);
      char* Bs = As + A3_ROWS * LDS_ROW * 2;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(KOD_ABL_NODMA)
#pragma unroll
      for (int i = 0; i < A3_PW; ++i) {
        const int blk = uwave + i * NW;
        if (blk < A3_INSTR) {
          const uint32_t vo = ((a3_mask[i] >> f_kh) & 1u) ? a3_voff[i] + soff : 0xFFFFFFF0u;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(As + blk * 16 * 64),
                                                   16, vo, 0, 0, 0);
        }
      }
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int q = 0; q < B_PER_WAVE; ++q) {
          const int blk = uwave * B_PER_WAVE + q;
          if (blk < B_INSTR)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(Bs + (kw * BN + blk * 16) * 64),
                                                     16, dbvoff[q], ((f_kh * 3 + kw) * a.cin_step + f_ci) * 2, 0, 0);
        }
#else
      (void)soff; (void)As; (void)Bs;
#endif
      f_ci += BK;
      if (f_ci >= a.cin_step) { f_ci = 0; ++f_kh; }
    };


    // ---- STEM staging state (see the header comment of this function)
    constexpr int AS_INSTR = AS_ROWS / 64;
    constexpr int AS_PW = (AS_INSTR + NW - 1) / NW;
    uint32_t as_voff[STEM ? AS_PW : 1], as_mask[STEM ? AS_PW : 1];
    if constexpr (STEM) {
#pragma unroll
      for (int i = 0; i < AS_PW; ++i) {
        const int blk = uwave + i * NW;               // DMA instruction = 64 staged rows of 16 B
        const int r = blk * 64 + lane;
        const long q = (long)m0 - 1 + r;              // output pixel whose CENTRE pair (tap j = 1) this row is
        const bool valid = blk < AS_INSTR && r < BM + 3 && q >= 0 && q < a.M;
        const int qq = valid ? (int)q : 0;
        int b, rem, oy, ox;
        fast_divmod(qq, HWo, a.rcp_hwo, b, rem);
        fast_divmod(rem, a.Wo, a.rcp_wo, oy, ox);
        const int by = oy * a.mul_h + a.add_h;
        as_voff[i] = (uint32_t)(((long)(b * a.Hs + by) * a.Ws + ox) * a.ldx * 2);
        uint32_t mk = 0;
        for (int kh = 0; kh < a.KH; ++kh)
          if ((unsigned)(by + kh) < (unsigned)a.Hs) mk |= 1u << kh;
        as_mask[i] = valid ? mk : 0u;
      }
#pragma unroll
      for (int jj = 0; jj < TM; ++jj) {
        const int p = m0 + wm * WM + jj * 32 + (lane & 31);
        int b, rem, oy, ox;
        fast_divmod(p < a.M ? p : 0, HWo, a.rcp_hwo, b, rem);
        fast_divmod(rem, a.Wo, a.rcp_wo, oy, ox);
        xmask |= (ox != 0 ? 1u : 0u) << (2 * jj) | (ox != a.Wo - 1 ? 1u : 0u) << (2 * jj + 1);
      }
    }
    auto dma_tileS = [&](int buf) {
      // step state: f_kh = kernel row
      const uint32_t soff = (uint32_t)(f_kh * a.Ws * a.ldx * 2);
      char* As = reinterpret_cast<char*>(lds + buf * STAGE_ELEMS);
      char* Bs = As + AS_ROWS * 16;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(KOD_ABL_NODMA)
#pragma unroll
      for (int i = 0; i < AS_PW; ++i) {
        const int blk = uwave + i * NW;
        if (blk < AS_INSTR) {
          const uint32_t vo = ((as_mask[i] >> f_kh) & 1u) ? as_voff[i] + soff : 0xFFFFFFF0u;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(As + blk * 1024), 16, vo, 0, 0, 0);
        }
      }
#pragma unroll
      for (int q = 0; q < B_PER_WAVE; ++q) {
        const int blk = uwave * B_PER_WAVE + q;
        if (blk < B_INSTR)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(Bs + blk * 16 * 64),
                                                   16, dbvoff[q], f_kh * (BK * 2), 0, 0);
      }
#else
      (void)soff; (void)As; (void)Bs;
#endif
      ++f_kh;
    };

    u32x4 areg[2];
    u32x4 breg[B_PER_THREAD];

    auto load_tile = [&](int kt) {
#ifdef KOD_ABL_NOGLOAD
      for (int r = 0; r < 2; ++r) areg[r] = u32x4{0u, 0u, 0u, 0u};
      for (int q = 0; q < B_PER_THREAD; ++q) breg[q] = u32x4{0u, 0u, 0u, 0u};
      return;
#endif
      const int k = kt * BK + a_chunk * 8;
      const uint32_t tap = __umulhi((uint32_t)k, a.magic_cin);
      const int ci = k - (int)tap * a.cin_step;
      const uint32_t kh = (a.KW == 1) ? tap : __umulhi(tap, a.magic_kw);
      const int kw = (int)tap - (int)kh * a.KW;
      const bool kvalid = k < a.K && ci < a.Cin;
      if (linear_taps) {
        const long toff = (long)(a.tap_sign * ((int)kh * a.Ws + kw)) * a.ldx + ci;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          u32x4 v = {0u, 0u, 0u, 0u};
          if (kvalid && ((rmask[r] >> tap) & 1u)) v = *reinterpret_cast<const u32x4*>(rptr[r] + toff);
          areg[r] = v;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          int ty = rby[r] + a.tap_sign * (int)kh;
          int tx = rbx[r] + a.tap_sign * kw;
          bool ok = rvalid[r] && kvalid && ty >= 0 && tx >= 0 &&
                    ((ty & ((1 << a.sh_shift) - 1)) == 0) && ((tx & ((1 << a.sw_shift) - 1)) == 0);
          int iy = ty >> a.sh_shift;
          int ix = tx >> a.sw_shift;
          ok = ok && iy < a.Hs && ix < a.Ws;
          u32x4 v = {0u, 0u, 0u, 0u};
          if (ok) {
            size_t off = (size_t)(rbase[r] + iy * a.Ws + ix) * a.ldx + a.xcoff + ci;
            v = *reinterpret_cast<const u32x4*>(a.x + off);
          }
          areg[r] = v;
        }
      }
#pragma unroll
      for (int q = 0; q < B_PER_THREAD; ++q) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (bptr[q]) v = *reinterpret_cast<const u32x4*>(bptr[q] + kt * BK);
        breg[q] = v;
      }
    };
    auto store_tile = [&](int buf) {
#ifdef KOD_ABL_NOLDSWRITE
      return;
#endif
      bf16_t* As = lds + buf * STAGE_ELEMS;
      bf16_t* Bs = As + BM * LDS_ROW;
#pragma unroll
      for (int r = 0; r < 2; ++r)
        *reinterpret_cast<u32x4*>(As + (a_row + r * 64) * LDS_ROW + a_chunk * 8) = areg[r];
#pragma unroll
      for (int q = 0; q < B_PER_THREAD; ++q) {
        int c = tid + q * NT;
        if (c < B_CHUNKS) *reinterpret_cast<u32x4*>(Bs + (c >> 2) * LDS_ROW + (c & 3) * 8) = breg[q];
      }
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int jj = 0; jj < TM; ++jj)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][jj][e] = 0.f;

    const int fr = lane & 31;
    const int fh = lane >> 5;
    auto compute = [&](int stage) {
      const bf16_t* As = lds + stage * STAGE_ELEMS;
      const bf16_t* Bs = As + BM * LDS_ROW;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 wf[TN], xf[TM];
#ifdef KOD_ABL_NODSREAD
        for (int i = 0; i < TN; ++i) { wf[i] = bf16x8{}; asm volatile("" : "+v"(wf[i])); }
        for (int jj = 0; jj < TM; ++jj) { xf[jj] = bf16x8{}; asm volatile("" : "+v"(xf[jj])); }
#else
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          int row = wn * WN + i * 32 + fr;
          int ch = FAST ? ((ks * 2 + fh) ^ ((row >> 2) & 3)) : (ks * 2 + fh);
          wf[i] = *reinterpret_cast<const bf16x8*>(Bs + row * LDS_ROW + ch * 8);
        }
#pragma unroll
        for (int jj = 0; jj < TM; ++jj) {
          int row = wm * WM + jj * 32 + fr;
          int ch = FAST ? ((ks * 2 + fh) ^ ((row >> 2) & 3)) : (ks * 2 + fh);
          xf[jj] = *reinterpret_cast<const bf16x8*>(As + row * LDS_ROW + ch * 8);
        }
#endif
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int jj = 0; jj < TM; ++jj)
#ifdef KOD_ABL_NOMFMA
            asm volatile("" :: "v"(wf[i]), "v"(xf[jj]));
#else
            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], xf[jj], acc[i][jj], 0, 0, 0);
#endif
      }
    };

    auto compute3 = [&](int stage) {
      const bf16_t* As = lds + stage * STAGE_ELEMS;
      const bf16_t* Bs = As + A3_ROWS * LDS_ROW;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        // output row `local` reads staged row local + 1 + tap_sign * (kw - 1)   (staged row 0 = pixel m0 - 1)
        const int roff = a.tap_sign > 0 ? kw : 2 - kw;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          bf16x8 wf[TN], xf[TM];
#pragma unroll
          for (int i = 0; i < TN; ++i) {
            const int row = wn * WN + i * 32 + fr;
            const int ch = (ks * 2 + fh) ^ ((row >> 2) & 3);
            wf[i] = *reinterpret_cast<const bf16x8*>(Bs + (kw * BN + row) * LDS_ROW + ch * 8);
          }
#pragma unroll
          for (int jj = 0; jj < TM; ++jj) {
            const int row = wm * WM + jj * 32 + fr + roff;
            const int ch = (ks * 2 + fh) ^ ((row >> 2) & 3);
            bf16x8 v = *reinterpret_cast<const bf16x8*>(As + row * LDS_ROW + ch * 8);
            if (kw == 0 && !((xmask >> (2 * jj)) & 1u)) v = bf16x8{};
            if (kw == 2 && !((xmask >> (2 * jj + 1)) & 1u)) v = bf16x8{};
            xf[jj] = v;
          }
#pragma unroll
          for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int jj = 0; jj < TM; ++jj)
              acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], xf[jj], acc[i][jj], 0, 0, 0);
        }
      }
    };

    auto computeS = [&](int stage) {
      const bf16_t* As = lds + stage * STAGE_ELEMS;
      const bf16_t* Bs = As + AS_ROWS * 8;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 wf[TN], xf[TM];
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          const int row = wn * WN + i * 32 + fr;
          const int ch = (ks * 2 + fh) ^ ((row >> 2) & 3);
          wf[i] = *reinterpret_cast<const bf16x8*>(Bs + row * LDS_ROW + ch * 8);
        }
#pragma unroll
        for (int jj = 0; jj < TM; ++jj) {
          // tap j = ks * 2 + fh of output pixel (local) i reads staged pair row i + j
          bf16x8 v = *reinterpret_cast<const bf16x8*>(As + (wm * WM + jj * 32 + fr + ks * 2 + fh) * 8);
          if (ks == 0 && fh == 0 && !((xmask >> (2 * jj)) & 1u)) v = bf16x8{};
          if (ks == 1 && fh == 0 && !((xmask >> (2 * jj + 1)) & 1u)) v = bf16x8{};
          xf[jj] = v;
        }
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int jj = 0; jj < TM; ++jj)
            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], xf[jj], acc[i][jj], 0, 0, 0);
      }
    };

    if constexpr (STEM) {
      dma_tileS(0);
      for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < nk) dma_tileS((kt + 1) & 1);
        computeS(kt & 1);
      }
      __syncthreads();
    } else if constexpr (ROW3) {
      dma_tile3(0);
      for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < nk) dma_tile3((kt + 1) & 1);
        compute3(kt & 1);
      }
      __syncthreads();
    } else if constexpr (FAST) {
      // LDS ring fed by LDS-DMA: while tile kt is multiplied, tiles kt+1 .. kt+NST-2 are in flight.  A wave's DMAs
      // retire in order, so "all but the newer tiles' instructions" is a counted vmcnt; the raw s_barrier then
      // publishes every wave's share of tile kt (and fences the stage that tile kt+NST-1 overwrites: it was last
      // read in step kt-1, which every wave has left once it reaches this barrier).
      const int my_b = (uwave * B_PER_WAVE < B_INSTR) ? ((B_INSTR - uwave * B_PER_WAVE) < B_PER_WAVE ? (B_INSTR - uwave * B_PER_WAVE) : B_PER_WAVE) : 0;
      dma_tile(0, 0);
      if (NST == 3 && nk > 1) dma_tile(1, 1);
      for (int kt = 0; kt < nk; ++kt) {
        if (NST == 3 && kt + 1 < nk) {
          // the newer tile's instructions of this wave (A_PER_WAVE pixel pieces + my_b weight pieces) may stay in flight
          if (my_b == 2) wait_vm_imm<A_PER_WAVE + 2>();
          else if (my_b == 1) wait_vm_imm<A_PER_WAVE + 1>();
          else wait_vm_imm<A_PER_WAVE>();
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (kt + NST - 1 < nk) dma_tile(kt + NST - 1, (kt + NST - 1) % NST);
#ifdef KOD_ABL_FIXUP
        // ablation only (tools/build_ablate.sh fixup -DKOD_ABL_FIXUP): what it would cost to apply the PRODUCER's BatchNorm +
        // SiLU in this consumer instead of in a separate pass - an in-LDS pass over the landed pixel tile (constant
        // scale / shift: the arithmetic, LDS traffic and extra barrier without even the parameter loads)
        {
          bf16_t* Af = lds + (kt % NST) * STAGE_ELEMS;
          for (int q = tid; q < BM * 4; q += NT) {
            bf16x8 v = *reinterpret_cast<bf16x8*>(Af + q * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float z = (float)v[e] * 1.001f + 0.01f;
              const float d = 1.0f + __expf(-z);
              float r = __builtin_amdgcn_rcpf(d);
              r = r * (2.0f - d * r);
              v[e] = (bf16_t)(z * r);
            }
            *reinterpret_cast<bf16x8*>(Af + q * 8) = v;
          }
          __builtin_amdgcn_s_barrier();
        }
#endif
        compute(kt % NST);
      }
      __syncthreads();
    } else {
      load_tile(0);
      store_tile(0);
      __syncthreads();
      for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        compute(buf);
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
      }
    }

    // ---- epilogue.  acc[i][jj][e]: pixel = wm*WM + jj*32 + (lane&31),
    //      channel = wn*WN + i*32 + 8*(e>>2) + 4*(lane>>5) + (e&3)
    if constexpr (MODE == MODE_HEAD) {
      // out[b][an][pix][slot] is, per anchor, ONE contiguous run over the tile's consecutive pixels: the tile goes
      // through LDS (64 rows per pass, fp32, padded rows) and leaves as lane-contiguous 4-byte stores (full lines per
      // instruction) instead of 60-byte-strided ones straight from the accumulators (1.6 -> 3+ TB/s on the 80x80 level).
      float* Cf = reinterpret_cast<float*>(lds);
      constexpr int CF_ROW = BN + 1;
      static_assert(64 * CF_ROW * 4 <= LDS_ELEMS * 2, "head epilogue staging must fit the operand ring");
      const int P = a.head_P, A_ = a.head_A;
      for (int h = 0; h < BM / 64; ++h) {
#pragma unroll
        for (int jj = 0; jj < TM; ++jj) {
          const int rt = wm * WM + jj * 32;                 // first tile row of this 32-row accumulator block
          if ((rt >> 6) == h) {
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
              for (int e = 0; e < 16; ++e) {
                const int col = wn * WN + i * 32 + 8 * (e >> 2) + 4 * fh + (e & 3);
                const int n = n0 + col;
                Cf[(rt - 64 * h + fr) * CF_ROW + col] = acc[i][jj][e] + (n < a.N ? a.bias[n] : 0.f);
              }
          }
        }
        __syncthreads();
        const int mh = m0 + 64 * h;
        for (int an = 0; an < A_; ++an)
          for (int idx = tid; idx < 64 * P; idx += NT) {
            const int r = (int)__umulhi((uint32_t)idx, a.magic_hp);     // idx / P (idx < 2^13: exact)
            const int slot = idx - r * P;
            const int m = mh + r;
            const int n = slot < 4 ? 4 * an + slot : (slot == 4 ? 4 * A_ + an : 5 * A_ + an * a.head_nc + slot - 5);
            if (m < a.M && n >= n0 && n < n0 + BN) {
              int b, pix;
              fast_divmod(m, HWo, a.rcp_hwo, b, pix);
              a.head_out[((size_t)(b * A_ + an) * HWo + pix) * P + slot] = Cf[r * CF_ROW + n - n0];
            }
          }
        __syncthreads();
      }
    } else {
      // F32ACC: own instantiations (a run-time branch here costs the default kernels 30-40 VGPRs and spills)
      if constexpr (F32ACC && (MODE == MODE_PLAIN || MODE == MODE_PLAIN_BN)) {
        if (a.f32_mode != 0) {
          // fp32 accumulation across producers, straight from / into the accumulators (lane: pixel fr of block jj,
          // channels 8g + 4fh .. +3 of block i): 16-byte accesses of the shadow
#pragma unroll
          for (int jj = 0; jj < TM; ++jj) {
            const int m = m0 + wm * WM + jj * 32 + fr;
            size_t opix = (size_t)m;
            if (a.out_mul != 1) {
              const int b = m / HWo, rem = m - b * HWo, oy = rem / a.Wo, ox = rem - oy * a.Wo;
              opix = ((size_t)b * a.out_H + oy * a.out_mul) * a.out_W + ox * a.out_mul;
            }
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * WN + i * 32 + 8 * g + 4 * fh;
                if (m >= a.M || n >= a.N) continue;
                int nch = n;
                size_t pix = opix;
                if (a.out_mul != 1) {
                  int poff = a.out_off_y * a.out_W + a.out_off_x;
                  if (a.d2s_C != 0) { const int cls = n / a.d2s_C; nch = n - cls * a.d2s_C; poff = (cls >> 1) * a.out_W + (cls & 1); }
                  pix += poff;
                }
                if (a.f32_mode == 4) {
                  const bf16x4 o = *reinterpret_cast<const bf16x4*>(a.y + pix * a.ldy + a.ycoff + nch);
#pragma unroll
                  for (int e = 0; e < 4; ++e) acc[i][jj][g * 4 + e] += (float)o[e];
                  continue;
                }
                f32x4* p32 = reinterpret_cast<f32x4*>(a.y32 + pix * a.ldy + a.ycoff + nch);
                if (a.f32_mode >= 2) {
                  const f32x4 o = *p32;
#pragma unroll
                  for (int e = 0; e < 4; ++e) acc[i][jj][g * 4 + e] += o[e];
                }
                if (a.f32_mode <= 2) {
                  f32x4 o;
#pragma unroll
                  for (int e = 0; e < 4; ++e) o[e] = acc[i][jj][g * 4 + e];
                  *p32 = o;
                }
              }
          }
        }
      }
      bf16_t* Cs = lds;
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int jj = 0; jj < TM; ++jj)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            bf16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (bf16_t)acc[i][jj][g * 4 + e];
            *reinterpret_cast<bf16x4*>(Cs + (wm * WM + jj * 32 + fr) * CS_ROW + wn * WN + i * 32 + 8 * g + 4 * fh) = v;
          }
      __syncthreads();
      constexpr int CPR = BN / 8;            // 16-byte chunks per row
      constexpr int RPP = NT / CPR;          // rows per pass
      const int c = tid % CPR;
      const int r0 = tid / CPR;
      const int n = n0 + c * 8;
      // folded stride-2 data gradient: this thread's 8 columns are 8 channels of ONE parity class (d2s_C % 8 == 0)
      int nch = n, pix_off = a.out_off_y * a.out_W + a.out_off_x;
      if constexpr (MODE == MODE_PLAIN || MODE == MODE_PLAIN_BN) {
        if (a.d2s_C != 0) {
          const int cls = n / a.d2s_C;
          nch = n - cls * a.d2s_C; pix_off = (cls >> 1) * a.out_W + (cls & 1);
        }
      }
      const bool col_ok = n < a.N;
      // MODE_PLAIN_BN: this thread's 8 channels belong to at most one segment (boundaries are multiples of 8)
      const bf16_t* sg_raw = nullptr;
      int sg_ldr = 0;
      float sg_sc[8], sg_sh[8];
      if constexpr (MODE == MODE_PLAIN_BN) {
#pragma unroll
        for (int sgi = 0; sgi < MAX_SEG; ++sgi)
          if (sgi < a.nseg && nch >= a.seg_begin[sgi] && nch < a.seg_end[sgi]) {
            const int cl = nch - a.seg_begin[sgi];
            sg_raw = a.seg_raw[sgi] + cl;
            sg_ldr = a.seg_ldr[sgi];
#pragma unroll
            for (int e = 0; e < 8; ++e) { sg_sc[e] = a.seg_aff[sgi][cl + e]; sg_sh[e] = a.seg_aff[sgi][a.seg_C[sgi] + cl + e]; }
          }
      }
      // stride-2 data gradients scatter their rows: (image, row, column) of this thread's first row by float-reciprocal
      // division, then advanced by RPP per pass (round 6: two integer divisions per pass - ~ 60 VALU instructions, 8 passes
      // per tile - in launches whose SIMDs are ~ 50 % VALU-busy, profiles/r06_s2_pmc.txt)
      int pb = 0, poy = 0, pox = 0;
      if constexpr (MODE == MODE_PLAIN || MODE == MODE_PLAIN_BN) {
        if (a.out_mul != 1) {
          int prem;
          fast_divmod(m0 + r0 < a.M ? m0 + r0 : 0, HWo, a.rcp_hwo, pb, prem);
          fast_divmod(prem, a.Wo, a.rcp_wo, poy, pox);
        }
      }
#pragma unroll
      for (int p = 0; p < BM / RPP; ++p) {
        int row = r0 + p * RPP;
        int m = m0 + row;
        if constexpr (MODE == MODE_PLAIN || MODE == MODE_PLAIN_BN) {
          if (p > 0 && a.out_mul != 1) {
            pox += RPP;
            while (pox >= a.Wo) { pox -= a.Wo; if (++poy == a.Ho) { poy = 0; ++pb; } }
          }
        }
        if (m < a.M && col_ok) {
          bf16x8 v = *reinterpret_cast<const bf16x8*>(Cs + row * CS_ROW + c * 8);
          size_t opix = (size_t)m;
          if constexpr (MODE == MODE_PLAIN || MODE == MODE_PLAIN_BN) {
            if (a.out_mul != 1)        // (pixel counts fit 31 bits: fill_common)
              opix = (size_t)(uint32_t)((pb * a.out_H + poy * a.out_mul) * a.out_W + pox * a.out_mul + pix_off);
          }
          bf16_t* dst = a.y + opix * a.ldy + a.ycoff + nch;
          if constexpr (MODE == MODE_PLAIN || MODE == MODE_PLAIN_BN) {
            if (a.accumulate && (!F32ACC || a.f32_mode < 2)) {
              bf16x8 o = *reinterpret_cast<const bf16x8*>(dst);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = (bf16_t)((float)v[e] + (float)o[e]);
            }
            if constexpr (MODE == MODE_PLAIN_BN) {
              if (sg_raw) {     // dz = dA * silu'(z), z = y*scale + shift, on the value as stored (bf16)
#ifdef KOD_ABL_NOBNLOAD       // (tools/build_ablate.sh: what the pre-BN re-read / the arithmetic / the stores of this epilogue cost)
                bf16x8 yv = v;
#else
                bf16x8 yv = *reinterpret_cast<const bf16x8*>(sg_raw + opix * sg_ldr);
#endif
#pragma unroll
                for (int e = 0; e < 8; ++e) {
#ifdef KOD_ABL_NOBNMATH
                  ssum[e] += (float)v[e]; ssq[e] += (float)yv[e]; continue;
#endif
                  // kodhip_common.h; the exponent's scale as one more multiply here: sixteen more per-channel constants
                  // would spill in the 128-register tiles
                  const float y = (float)yv[e];
                  const float z = __builtin_fmaf(y, sg_sc[e], sg_sh[e]);
                  const float r = kod_sigmoid_l2(KOD_NEG_LOG2E * z);
                  const float dz = kod_silu_bwd((float)v[e], z, r);
                  ssum[e] += dz;
                  ssq[e] = __builtin_fmaf(dz, y, ssq[e]);
                }
              }
            }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float f = (float)v[e];
              ssum[e] += f;
              ssq[e] += f * f;
            }
          }
#ifdef KOD_ABL_NOSTORE
          if (v[0] == (bf16_t)12345.f)
#endif
          *reinterpret_cast<bf16x8*>(dst) = v;
        }
      }
      __syncthreads();
      if constexpr (MODE == MODE_PLAIN_BN) {
        // fold this tile's sums into one running value per (channel, statistic), held by thread tid < 2*BN: the
        // 16 per-thread accumulators then live only inside the epilogue, where the MFMA accumulators are dead
        // (fixed order: rows by shuffle, waves 0..NW-1 by the owner thread => deterministic)
#pragma unroll
        for (int o = CPR; o < 64; o <<= 1)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            ssum[e] += __shfl_xor(ssum[e], o, 64);
            ssq[e] += __shfl_xor(ssq[e], o, 64);
          }
        if (lane < CPR) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            sred[(wave * BN + lane * 8 + e) * 2 + 0] = ssum[e];
            sred[(wave * BN + lane * 8 + e) * 2 + 1] = ssq[e];
          }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) ssum[e] = ssq[e] = 0.f;
        __syncthreads();
        if (tid < BN * 2) {
          const int ch = tid >> 1, st = tid & 1;
#pragma unroll
          for (int w = 0; w < NW; ++w) bn_run += sred[(w * BN + ch) * 2 + st];
        }
        __syncthreads();      // the next tile's DMA overwrites sred
      }
    }
  }

  if constexpr (MODE == MODE_PLAIN_BN) {
    if (tid < BN * 2) {
      const int ch = tid >> 1, st = tid & 1;
      int nn = n0 + ch, sbase = a.slot_base;
      if (a.d2s_C != 0) {       // folded stride-2 form: the four parity classes of a channel reduce into disjoint slot ranges
        const int cls = nn / a.d2s_C;
        nn -= cls * a.d2s_C; sbase = cls * a.groups_m;
        if (cls > 3) nn = -1;
      }
#pragma unroll
      for (int sgi = 0; sgi < MAX_SEG; ++sgi)
        if (sgi < a.nseg && nn >= a.seg_begin[sgi] && nn < a.seg_end[sgi]) {
          float* slot = a.seg_part[sgi] + ((size_t)st * a.seg_C[sgi] + (nn - a.seg_begin[sgi])) * a.stats_slots;
          slot[sbase + gm] = bn_run;
          for (int t = sbase + gm + a.slot_used; t < a.stats_slots; t += a.slot_used) slot[t] = 0.f;
        }
    }
  }
  if constexpr (MODE == MODE_RAW) {
    constexpr int CPR = BN / 8;
#pragma unroll
    for (int o = CPR; o < 64; o <<= 1)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        ssum[e] += __shfl_xor(ssum[e], o, 64);
        ssq[e] += __shfl_xor(ssq[e], o, 64);
      }
    if (lane < CPR) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        sred[(wave * BN + lane * 8 + e) * 2 + 0] = ssum[e];
        sred[(wave * BN + lane * 8 + e) * 2 + 1] = ssq[e];
      }
    }
    __syncthreads();
    if (tid < BN * 2) {
      int ch = tid >> 1, st = tid & 1;
      // lanes of different waves cover the same chunk set only when CPR divides the wave evenly:
      // thread t handles chunk t % CPR, so every wave holds every chunk (CPR <= 64).
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += sred[(w * BN + ch) * 2 + st];
      int nn = n0 + ch;
      if (nn < a.N) {
        float* slot = a.stats + ((size_t)st * a.N + nn) * a.stats_slots;
        slot[gm] = s;
        for (int t = gm + a.groups_m; t < a.stats_slots; t += a.groups_m) slot[t] = 0.f;   // slots this launch does not use
      }
    }
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int MODE, bool FAST, bool F32ACC = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, (MODE == 2 /*HEAD*/ ? 2 : (FAST ? 4 : 3)))   /* waves per SIMD */
void conv_igemm_kernel(ConvArgs a) {
  conv_igemm_body<BM, BN, WAVES_M, WAVES_N, MODE, FAST, false, F32ACC>(a, blockIdx.x);
}

template <int BN, int WAVES_M, int WAVES_N, int MODE, bool F32ACC = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, (BN == 128 ? 2 : (BN == 64 ? 3 : 4)))    /* = what the LDS stages allow */
void conv_igemm_row3_kernel(ConvArgs a) {
  conv_igemm_body<128, BN, WAVES_M, WAVES_N, MODE, true, true, F32ACC>(a, blockIdx.x);
}

template <int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, 4)
void conv_igemm_stem_kernel(ConvArgs a) {
  conv_igemm_body<128, BN, WAVES_M, WAVES_N, MODE_RAW, true, false, false, true>(a, blockIdx.x);
}

// ---- the stem as its own kernel: weights resident, whole-tile double buffering -------------------------------------
// conv_igemm_body<STEM> cut the stem's fill to a quarter but still ran one barrier + one DMA round trip per kernel row
// (two MFMAs per wave between them): 290 us for a layer whose traffic is 0.52 GB.  Here the block keeps ALL weights of the
// stem in LDS for its lifetime (6 kernel rows x 32 channels x 32 k = 12 KB), stages the six pair-row runs of a WHOLE
// 128-pixel tile at once (6 x 3 KB) and does so for tile t + 1 while tile t is multiplied and stored: one counted wait
// per tile, DMA latency behind a full tile of work.  Measured at B = 64 / 640 px: generic per-tap staging 355 us, row-shared
// staging in the generic body 290 us, this kernel 240 us (0.52 GB of traffic: 2.2 TB/s; what is left is the prefetch depth
// of one tile per block, three blocks per CU).  N <= 32.
__global__ __launch_bounds__(256, 3) void conv_stem_fwd_kernel(ConvArgs a) {
  constexpr int BM = 128, BN = 32, NWV = 4;
  constexpr int PR = 192;                              // staged pair rows per kernel row (BM + 3 used)
  constexpr int KHM = 6;
  constexpr int W_BYTES = KHM * BN * 64;               // 12 KB
  constexpr int P_BYTES = KHM * PR * 16;               // 18 KB per tile
  constexpr int CS_ROW = BN + 8;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[W_BYTES + 2 * P_BYTES];
  float* sred = reinterpret_cast<float*>(lds + W_BYTES);      // reused after the tile loop

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, fh = lane >> 5;
  const int HWo = a.Ho * a.Wo;
  const int grid = gridDim.x, blk = blockIdx.x;

#if defined(__HIP_DEVICE_COMPILE__)
  __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);
#endif
  // ---- weights, once: instruction t = wave + 4 i (12 of 1 KB): kernel row t / 2, rows 16 (t & 1) .. + 16
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int t = wave + NWV * i;
    const int kh = t >> 1, row = (t & 1) * 16 + (lane >> 2);
    const int chunk = (lane & 3) ^ ((row >> 2) & 3);
    const uint32_t vo = row < a.N ? (uint32_t)(((size_t)row * a.Kp + kh * 32 + chunk * 8) * 2) : 0xFFFFFFF0u;
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(lds + t * 1024), 16, vo, 0, 0, 0);
#else
    (void)vo;
#endif
  }
  // ---- pixel rows of one tile: instruction t = wave + 4 i (18 of 1 KB): kernel row t / 3, rows 64 (t % 3) .. + 64
  auto stage_tile = [&](int mt, int buf) {
    const int m0 = mt * BM;
    unsigned char* P = lds + W_BYTES + buf * P_BYTES;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int t = wave + NWV * i;
      if (t >= 18) continue;
      const int kh = t / 3, part = t - kh * 3;
      const int r = part * 64 + lane;
      const long q = (long)m0 - 1 + r;               // output pixel whose centre pair this row is
      bool ok = r < BM + 3 && q >= 0 && q < a.M;
      int b, rem, oy, ox;
      fast_divmod(ok ? (int)q : 0, HWo, a.rcp_hwo, b, rem);
      fast_divmod(rem, a.Wo, a.rcp_wo, oy, ox);
      const int iy = oy * a.mul_h + a.add_h + kh;
      ok = ok && (unsigned)iy < (unsigned)a.Hs;
      const uint32_t vo = ok ? (uint32_t)(((long)(b * a.Hs + iy) * a.Ws + ox) * 16) : 0xFFFFFFF0u;
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(P + t * 1024), 16, vo, 0, 0, 0);
#else
      (void)vo; (void)P;
#endif
    }
  };

  float ssum[8], ssq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) ssum[e] = ssq[e] = 0.f;

  const int my_p = wave < 2 ? 5 : 4;                   // this wave's share of a tile's 18 DMA instructions
  int buf = 0;
  // tiles in bands per XCD (see conv_igemm_body): vertically adjacent tiles share four of their six input rows
  const int xcd = blk & 7, bq = blk >> 3, bk = (grid - xcd + 7) >> 3;
  int bbefore = 0;
  for (int i = 0; i < xcd; ++i) bbefore += (grid - i + 7) >> 3;
  const int band_end = (int)((long)a.tiles_m * (bbefore + bk) / grid);
  const int mt0 = (int)((long)a.tiles_m * bbefore / grid) + bq;
  if (mt0 < band_end) stage_tile(mt0, 0);
  for (int mt = mt0; mt < band_end; mt += bk) {
    const int m0 = mt * BM;
    const bool more = mt + bk < band_end;
    if (more) stage_tile(mt + bk, buf ^ 1);
    // everything but the next tile's instructions of this wave must have landed (the weights too, first time round).  The
    // previous tile's stores may still be outstanding and may retire in any order relative to the loads - that only makes
    // the wait longer: DMA loads retire in order among themselves, so the count cannot fall to my_p while one of this
    // tile's loads is pending
    if (more) { if (my_p == 5) wait_vm_imm<5>(); else wait_vm_imm<4>(); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // this lane's output pixel: taps that would read the neighbouring image row's pairs are zero padding
    int pb, prem, poy, pox;
    const int pm = m0 + wave * 32 + fr;
    fast_divmod(pm < a.M ? pm : 0, HWo, a.rcp_hwo, pb, prem);
    fast_divmod(prem, a.Wo, a.rcp_wo, poy, pox);
    const bool ok0 = pox != 0, ok2 = pox != a.Wo - 1;

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const bf16_t* Wl = reinterpret_cast<const bf16_t*>(lds);
    const bf16_t* Pl = reinterpret_cast<const bf16_t*>(lds + W_BYTES + buf * P_BYTES);
#pragma unroll
    for (int kh = 0; kh < KHM; ++kh)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int ch = (ks * 2 + fh) ^ ((fr >> 2) & 3);
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(Wl + (kh * BN + fr) * 32 + ch * 8);
        bf16x8 xf = *reinterpret_cast<const bf16x8*>(Pl + (kh * PR + wave * 32 + fr + ks * 2 + fh) * 8);
        if (ks == 0 && fh == 0 && !ok0) xf = bf16x8{};
        if (ks == 1 && fh == 0 && !ok2) xf = bf16x8{};
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc, 0, 0, 0);
      }
    __builtin_amdgcn_s_barrier();                      // every wave has read this tile's rows: its buffer becomes the store staging

    // ---- epilogue: acc[e]: pixel fr of this wave, channel 8 (e >> 2) + 4 fh + (e & 3); through LDS so that the tile leaves
    //      as 16-byte channel-contiguous stores (measured: 240 us; 8-byte stores straight from the accumulators, one barrier
    //      per tile, statistics kept per lane: 257 us)
    bf16_t* Cs = reinterpret_cast<bf16_t*>(lds + W_BYTES + buf * P_BYTES);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (bf16_t)acc[g * 4 + e];
      *reinterpret_cast<bf16x4*>(Cs + (wave * 32 + fr) * CS_ROW + 8 * g + 4 * fh) = v;
    }
    __syncthreads();
    const int c = tid & 3, r0 = tid >> 2;
    const bool col_ok = c * 8 < a.N;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = r0 + p * 64;
      const int m = m0 + row;
      if (m < a.M && col_ok) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(Cs + row * CS_ROW + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float f = (float)v[e];
          ssum[e] += f;
          ssq[e] += f * f;
        }
        *reinterpret_cast<bf16x8*>(a.y + (size_t)m * a.ldy + a.ycoff + c * 8) = v;
      }
    }
    __syncthreads();                                   // the next iteration's DMA overwrites this buffer
    buf ^= 1;
  }

  // ---- BatchNorm partial statistics of this block: rows by shuffle, waves by the owner thread (fixed order)
#pragma unroll
  for (int o = 4; o < 64; o <<= 1)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      ssum[e] += __shfl_xor(ssum[e], o, 64);
      ssq[e] += __shfl_xor(ssq[e], o, 64);
    }
  if (lane < 4) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sred[(wave * BN + lane * 8 + e) * 2 + 0] = ssum[e];
      sred[(wave * BN + lane * 8 + e) * 2 + 1] = ssq[e];
    }
  }
  __syncthreads();
  if (tid < BN * 2) {
    const int ch = tid >> 1, st = tid & 1;
    float sacc = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) sacc += sred[(w * BN + ch) * 2 + st];
    if (ch < a.N) {
      float* slot = a.stats + ((size_t)st * a.N + ch) * a.stats_slots;
      slot[blk] = sacc;
      for (int t = blk + grid; t < a.stats_slots; t += grid) slot[t] = 0.f;     // slots this launch does not use
    }
  }
}

// Four problems of identical tiling in one launch (the parity classes of a stride-2 dgrad).  The classes have 4, 2, 2
// and 1 taps - reductions of very different length - and the launch is a few rounds of resident blocks at most, so
// blocks are dispatched longest class first (class 3, then 1 and 2, then 0): a long block never starts in the last
// round behind short ones.  (The layers that still take this form are the deep ones, whose dY stays in L2 / MALL
// whatever the order; `interleave` != 0 restores the round-1 order - blocks 8j .. 8j+31 = {XCD 0-7} x {class 0-3} -
// for A/B.)
struct ConvArgs4 { ConvArgs c[4]; int per_class, interleave; };

template <int BM, int BN, int WAVES_M, int WAVES_N, int MODE, bool FAST, bool F32ACC = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, (MODE == 2 /*HEAD*/ ? 2 : (FAST ? 4 : 3)))   /* waves per SIMD */
void conv_igemm_x4_kernel(ConvArgs4 p) {
  const int bid = blockIdx.x;
  if (p.interleave) {
    const int cls = (bid >> 3) & 3;
    conv_igemm_body<BM, BN, WAVES_M, WAVES_N, MODE, FAST, false, F32ACC>(p.c[cls], ((bid >> 5) << 3) | (bid & 7));
  } else {
    const int q = bid / p.per_class;                     // 0..3 in dispatch order
    const int cls = q == 0 ? 3 : (q == 3 ? 0 : q);
    conv_igemm_body<BM, BN, WAVES_M, WAVES_N, MODE, FAST, false, F32ACC>(p.c[cls], bid - q * p.per_class);
  }
}

// ---- launch planning --------------------------------------------------------------------------------
// Resident blocks per chip (256 CUs) for each tile shape (FAST path: register-bound at 4 x 4 waves or 2 x 8 waves
// per CU; register-staged path: LDS-bound, 2 stages of (128+BN)*80 B).
struct Plan { int bm, bn, groups_m, tiles_m, tiles_n, grid; };

int slots_for(int bm, int bn, bool fast) {
  if (bm == 256) return 512;
  if (fast) return 1024;                        // 4 blocks per CU (128-VGPR launch bound)
  return bn == 128 ? 1024 : (bn == 64 ? 1280 : 1536);
}

constexpr int MAX_STATS_SLOTS = 1024;


// Tile shape: the widest channel tile that fits N (or the next narrower one when that removes a badly quantised
// last round).  256-pixel tiles (8 waves, 3-stage ring) when the reduction is long enough to amortise their deeper
// pipeline fill (measured on gfx950: 3x3 layers with N >= 128 gain 15-25 %, short-K 1x1 layers lose ~5 %).
Plan make_plan(long M, int N, int K, bool fast, bool row3 = false, int bn_cap = 0) {
  static int force_bn = -1, force_bm = -1;
  if (force_bn < 0) { const char* e = getenv("KODHIP_FORCE_BN"); force_bn = e ? atoi(e) : 0; }
  if (force_bm < 0) { const char* e = getenv("KODHIP_FORCE_BM"); force_bm = e ? atoi(e) : 0; }
  // 64 < N < 128 (yv5m's 96): one 128-column tile (three quarters used) stages every pixel tile once, two 64-column tiles twice
  static int wide96 = -1;                  // KODHIP_PLAN_WIDE96=0: A/B knob
  if (wide96 < 0) { const char* e = getenv("KODHIP_PLAN_WIDE96"); wide96 = e ? atoi(e) : 1; }
  int widest = (N >= 128 || (wide96 && N > 64)) ? 128 : (N > 32 ? 64 : 32);
  if (bn_cap && widest > bn_cap) widest = bn_cap;
  const bool can256 = fast && widest >= 64 && M >= 256 * 64;       // 256 x 128 and 256 x 64 tiles
  const int bm = !row3 && can256 && (force_bm ? force_bm == 256 : K >= 512) ? 256 : 128;
  Plan best = {};
  double best_cost = 1e30;
  // (128 < N <= 192 as three 64-column tiles instead of two 128-column ones: 3x3 + 5 %, stride-2 forward - 39 %, 1x1 - 10 %: not taken)
  for (int bn = widest; bn >= 32 && bn >= widest / 2; bn >>= 1) {
    if (bm == 256 && bn != widest) continue;
    if (force_bn && force_bn <= widest && bn != force_bn) continue;
    long tiles_m = (M + bm - 1) / bm, tiles_n = (N + bn - 1) / bn;
    int slots = slots_for(bm, bn, fast);
    if (row3) slots = bn == 128 ? 512 : (bn == 64 ? 768 : 1024);       // ROW3 stages are larger: 2 / 3 / 4 blocks per CU
    long rounds = (tiles_m * tiles_n + slots - 1) / slots;
    double cost = (double)rounds * (bm + bn);       // per-round time ~ operand bytes staged per tile
    if (cost < best_cost * 0.97) {
      best_cost = cost;
      int target = slots / (int)tiles_n;
      if (target < 8) target = 8;
      if (target > MAX_STATS_SLOTS) target = MAX_STATS_SLOTS;
      int gm = tiles_m < target ? (int)tiles_m : target;
      best = {bm, bn, gm, (int)tiles_m, (int)tiles_n, cdiv(gm, 8) * 8 * (int)tiles_n};
    }
  }
  return best;
}

bool fast_eligible(const ConvArgs& a);

// ROW3 form (conv_igemm_body): 3x3 / stride 1 / pad 1 gathers of same-size images on the FAST path.
// KODHIP_ROW3: 0 = off, 1 = only where the plan would use 128-pixel tiles anyway, 2 (default) = every eligible layer.
bool row3_eligible(const ConvArgs& a, bool fast) {
  static int mode = -1;
  if (mode < 0) { const char* e = getenv("KODHIP_ROW3"); mode = e ? atoi(e) : 2; }
  if (!mode || !fast) return false;
  const bool shape = a.KH == 3 && a.KW == 3 && a.mul_h == 1 && a.mul_w == 1 && a.Ho == a.Hs && a.Wo == a.Ws &&
                     a.add_h == -a.tap_sign && a.add_w == -a.tap_sign && a.wide_px == 1 && a.nk1 == 0 && a.d2s_C == 0 &&
                     a.out_mul == 1 && a.head_out == nullptr;
  if (!shape) return false;
  if (mode == 1 && make_plan(a.M, a.N, a.K, fast).bm != 128) return false;
  return true;
}

// folded stride-2 data gradient: d2s_skip pays where the class boundary 2 * Cin falls on a tile boundary of the plan the
// layer has anyway (Cin = 64: 170.7 -> 153.0 us); forcing 64-column tiles on Cin = 32 to get one costs more in
// re-staging than the skipped taps save (236 -> 263 us), so no cap is applied (KODHIP_S2F_BN_CAP=64: the experiment)
int fold_bn_cap(const ConvArgs& a) {
  static int cap = -1;
  if (cap < 0) { const char* e = getenv("KODHIP_S2F_BN_CAP"); cap = e ? atoi(e) : 0; }
  return (cap && a.d2s_C != 0 && a.d2s_skip && 2 * a.d2s_C == cap) ? cap : 0;
}

Plan plan_conv(const ConvArgs& a, bool fast, bool& row3, int mode) {
  static int modes = -1;                 // KODHIP_ROW3_MODES: bit per MODE (debug: 1 = forward, 2 = dgrad, 8 = dgrad + BN reduction)
  if (modes < 0) { const char* e = getenv("KODHIP_ROW3_MODES"); modes = e ? atoi(e) : 0xF; }
  row3 = row3_eligible(a, fast) && ((modes >> mode) & 1);
  return make_plan(a.M, a.N, a.nk1 ? 2 * a.K : a.K, fast, row3, fold_bn_cap(a));
}

template <int MODE, bool F32ACC = false>
int launch(const ConvArgs& a, hipStream_t stream) {
  if constexpr ((MODE == MODE_PLAIN || MODE == MODE_PLAIN_BN) && !F32ACC) {
    if (a.f32_mode != 0) return launch<MODE, true>(a, stream);      // fp32 accumulation across producers: own kernels
  }
  ConvArgs args = a;
  const long xb = (long)a.B * a.Hs * a.Ws * a.ldx * 2, wb = (long)a.N * a.Kp * 2;
  const bool fast = fast_eligible(a);
  KOD_CHECK_ARG(fast || a.wide_px == 1, "conv: wide-pixel taps need the FAST path");
  args.x_bytes = (uint32_t)xb; args.w_bytes = (uint32_t)wb;
  static const bool pw_on = !(getenv("KODHIP_POINTWISE") && getenv("KODHIP_POINTWISE")[0] == '0');     // A/B knob
  args.pointwise = (pw_on && fast && a.KH == 1 && a.KW == 1 && a.mul_h == 1 && a.mul_w == 1 && a.add_h == 0 && a.add_w == 0 &&
                    a.Ho == a.Hs && a.Wo == a.Ws && a.wide_px == 1 && (a.sh_shift | a.sw_shift) == 0) ? 1 : 0;
  bool row3;
  const Plan p = plan_conv(a, fast, row3, MODE);
  args.tiles_n = p.tiles_n; args.tiles_m = p.tiles_m; args.groups_m = p.groups_m;
  if (MODE == MODE_RAW) {
    KOD_CHECK_ARG(a.stats_slots >= p.groups_m, "conv: stats buffer has %d slots, launch needs %d", a.stats_slots, p.groups_m);
  }
  if (MODE == MODE_PLAIN_BN) {
    const int need = (a.d2s_C ? 4 : 1) * p.groups_m;
    KOD_CHECK_ARG(a.stats_slots >= need, "conv: partial buffers have %d slots, launch needs %d", a.stats_slots, need);
    args.slot_base = 0; args.slot_used = need;
  }
  dim3 g(p.grid);
  if constexpr (MODE == MODE_RAW) {
    // the stem's wide-pixel form (KODHIP_STEM_ROW=0: the generic per-tap staging, for A/B)
    static const bool stem_row = !(getenv("KODHIP_STEM_ROW") && getenv("KODHIP_STEM_ROW")[0] == '0');
    if (stem_row && fast && !row3 && a.wide_px == 4 && a.KW == 1 && a.mul_w == 1 && a.add_w == -1 && a.ldx == 8 && a.xcoff == 0 &&
        p.bm == 128 && (p.bn == 32 || p.bn == 64) && (long)a.M + 256 < (1l << 31)) {
      // KODHIP_STEM_ROW: 1 = row-shared staging inside the generic body, default = the dedicated kernel (N <= 32)
      if (p.bn == 32 && a.N <= 32 && a.KH == 6 && a.ycoff % 8 == 0 && !(getenv("KODHIP_STEM_ROW") && getenv("KODHIP_STEM_ROW")[0] == '1')) {
        int blocks = p.tiles_m < 768 ? p.tiles_m : 768;             // 3 resident blocks per CU (48 KB of LDS each)
        if (blocks > a.stats_slots) blocks = a.stats_slots;
        hipLaunchKernelGGL(conv_stem_fwd_kernel, dim3(blocks), dim3(256), 0, stream, args);
        KOD_LAUNCH_CHECK("conv_stem_fwd");
        return KOD_OK;
      }
      if (p.bn == 32) hipLaunchKernelGGL((conv_igemm_stem_kernel<32, 4, 1>), g, dim3(256), 0, stream, args);
      else hipLaunchKernelGGL((conv_igemm_stem_kernel<64, 2, 2>), g, dim3(256), 0, stream, args);
      KOD_LAUNCH_CHECK("conv_igemm_stem");
      return KOD_OK;
    }
  }
  if constexpr (MODE != MODE_HEAD) {
    if (row3) {
      if (p.bn == 128) hipLaunchKernelGGL((conv_igemm_row3_kernel<128, 2, 2, MODE, F32ACC>), g, dim3(256), 0, stream, args);
      else if (p.bn == 64) hipLaunchKernelGGL((conv_igemm_row3_kernel<64, 2, 2, MODE, F32ACC>), g, dim3(256), 0, stream, args);
      else hipLaunchKernelGGL((conv_igemm_row3_kernel<32, 4, 1, MODE, F32ACC>), g, dim3(256), 0, stream, args);
      KOD_LAUNCH_CHECK("conv_igemm_row3");
      return KOD_OK;
    }
  }
  if (fast) {
    if (p.bm == 256 && p.bn == 64) hipLaunchKernelGGL((conv_igemm_kernel<256, 64, 4, 2, MODE, true, F32ACC>), g, dim3(512), 0, stream, args);
    else if (p.bm == 256) hipLaunchKernelGGL((conv_igemm_kernel<256, 128, 4, 2, MODE, true, F32ACC>), g, dim3(512), 0, stream, args);
    else if (p.bn == 128) hipLaunchKernelGGL((conv_igemm_kernel<128, 128, 2, 2, MODE, true, F32ACC>), g, dim3(256), 0, stream, args);
    else if (p.bn == 64) hipLaunchKernelGGL((conv_igemm_kernel<128, 64, 2, 2, MODE, true, F32ACC>), g, dim3(256), 0, stream, args);
    else hipLaunchKernelGGL((conv_igemm_kernel<128, 32, 4, 1, MODE, true, F32ACC>), g, dim3(256), 0, stream, args);
  } else if constexpr (F32ACC) {
    KOD_CHECK_ARG(false, "conv: fp32 accumulation across producers needs the FAST path");
  } else if constexpr (MODE == MODE_PLAIN_BN) {
    KOD_CHECK_ARG(false, "conv: the fused BatchNorm-backward reduction needs the FAST path (query the slots first)");
  } else {
    if (p.bn == 128) hipLaunchKernelGGL((conv_igemm_kernel<128, 128, 2, 2, MODE, false>), g, dim3(256), 0, stream, args);
    else if (p.bn == 64) hipLaunchKernelGGL((conv_igemm_kernel<128, 64, 2, 2, MODE, false>), g, dim3(256), 0, stream, args);
    else hipLaunchKernelGGL((conv_igemm_kernel<128, 32, 4, 1, MODE, false>), g, dim3(256), 0, stream, args);
  }
  KOD_LAUNCH_CHECK("conv_igemm");
  return KOD_OK;
}

// one launch for four FAST problems that share M, N and the tile plan (chosen for the longest reduction)
template <int MODE, bool F32ACC = false>
int launch_x4(ConvArgs c[4], hipStream_t stream) {
  if constexpr (!F32ACC) {
    if (c[0].f32_mode != 0) return launch_x4<MODE, true>(c, stream);
  }
  ConvArgs4 p;
  int kmax = 0;
  for (int i = 0; i < 4; ++i) kmax = c[i].K > kmax ? c[i].K : kmax;
  const Plan pl = make_plan(c[0].M, c[0].N, kmax, true);
  for (int i = 0; i < 4; ++i) {
    p.c[i] = c[i];
    p.c[i].x_bytes = (uint32_t)((long)c[i].B * c[i].Hs * c[i].Ws * c[i].ldx * 2);
    p.c[i].w_bytes = (uint32_t)((long)c[i].N * c[i].Kp * 2);
    p.c[i].tiles_n = pl.tiles_n; p.c[i].tiles_m = pl.tiles_m; p.c[i].groups_m = pl.groups_m;
    if (MODE == MODE_PLAIN_BN) {      // the four classes reduce into disjoint slot ranges of the same buffers
      KOD_CHECK_ARG(c[i].stats_slots >= 4 * pl.groups_m, "conv: partial buffers have %d slots, launch needs %d", c[i].stats_slots, 4 * pl.groups_m);
      p.c[i].slot_base = i * pl.groups_m; p.c[i].slot_used = 4 * pl.groups_m;
    }
  }
  static const bool interleave = getenv("KODHIP_S2_INTERLEAVE") != nullptr;
  p.per_class = pl.grid; p.interleave = interleave;
  dim3 g(pl.grid * 4);
  if (pl.bm == 256 && pl.bn == 64) hipLaunchKernelGGL((conv_igemm_x4_kernel<256, 64, 4, 2, MODE, true, F32ACC>), g, dim3(512), 0, stream, p);
  else if (pl.bm == 256) hipLaunchKernelGGL((conv_igemm_x4_kernel<256, 128, 4, 2, MODE, true, F32ACC>), g, dim3(512), 0, stream, p);
  else if (pl.bn == 128) hipLaunchKernelGGL((conv_igemm_x4_kernel<128, 128, 2, 2, MODE, true, F32ACC>), g, dim3(256), 0, stream, p);
  else if (pl.bn == 64) hipLaunchKernelGGL((conv_igemm_x4_kernel<128, 64, 2, 2, MODE, true, F32ACC>), g, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL((conv_igemm_x4_kernel<128, 32, 4, 1, MODE, true, F32ACC>), g, dim3(256), 0, stream, p);
  KOD_LAUNCH_CHECK("conv_igemm_x4");
  return KOD_OK;
}

bool fast_eligible(const ConvArgs& a) {
  const long xb = (long)a.B * a.Hs * a.Ws * a.ldx * 2, wb = (long)a.N * a.Kp * 2;
  return (a.cin_step % 32 == 0) && (a.sh_shift | a.sw_shift) == 0 && a.K == a.Kp && xb < (1l << 32) - 64 &&
         wb < (1l << 32) - 64 && !getenv("KODHIP_NO_FAST");
}

int ilog2_exact(int v) { int s = 0; while ((1 << s) < v) ++s; return ((1 << s) == v) ? s : -1; }

}  // namespace

extern "C" {

// Per-channel partial slots of the statistics buffer: it must hold 2 * N * kodhip_conv_stats_slots(M, N) floats;
// the kernel fills the slots it uses and zeroes the rest.
int kodhip_conv_stats_slots(long M, int N) {
  long tiles_m = (M + 127) / 128;                  // upper bound over every tile shape the launcher may pick
  (void)N;
  return tiles_m < MAX_STATS_SLOTS ? (int)tiles_m : MAX_STATS_SLOTS;
}

static int fill_common(ConvArgs& a, const void* x, const void* w, int B, int Hs, int Ws, int ldx, int xcoff,
                       int Cin, int Ho, int Wo, int N, int KH, int KW, int Kp) {
  KOD_CHECK_ARG(x && w, "conv: null pointer");
  KOD_CHECK_ARG(B > 0 && Hs > 0 && Ws > 0 && Ho > 0 && Wo > 0 && N > 0, "conv: bad dims");
  KOD_CHECK_ARG(Cin % 8 == 0 && ldx % 8 == 0 && xcoff % 8 == 0, "conv: channels must be multiples of 8 (Cin=%d ldx=%d off=%d)", Cin, ldx, xcoff);
  // a tap may span several consecutive pixels (Cin = wide * ldx, one column of taps): the stem's pixel-pair window
  const bool wide = Cin > ldx && xcoff == 0 && Cin % ldx == 0 && KW == 1 && Cin == 32;
  KOD_CHECK_ARG(xcoff + Cin <= ldx || wide, "conv: channel slice out of range");
  a.wide_px = wide ? Cin / ldx : 1;
  const int cin_step = (Cin + 31) / 32 * 32;
  KOD_CHECK_ARG(Kp == KH * KW * cin_step, "conv: Kp=%d must be taps x round_up(Cin, 32) = %d (packed K axis: k = tap * round_up(Cin, 32) + ci)", Kp, KH * KW * cin_step);
  KOD_CHECK_ARG(KH * KW <= 32, "conv: at most 32 taps");
  KOD_CHECK_ARG((long)B * Hs * Ws < (1l << 31) / 1 && (long)B * Ho * Wo < (1l << 31), "conv: pixel count overflows int32");
  a.x = (const bf16_t*)x; a.w = (const bf16_t*)w;
  a.B = B; a.Hs = Hs; a.Ws = Ws; a.ldx = ldx; a.xcoff = xcoff; a.Cin = Cin; a.cin_step = cin_step;
  a.Ho = Ho; a.Wo = Wo; a.M = B * Ho * Wo; a.N = N; a.K = KH * KW * cin_step; a.Kp = Kp; a.KH = KH; a.KW = KW;
  a.magic_cin = magic_u32((uint32_t)cin_step); a.magic_kw = magic_u32((uint32_t)KW);
  a.out_mul = 1;
  a.rcp_hwo = 1.0f / (float)(Ho * Wo); a.rcp_wo = 1.0f / (float)Wo;
  return KOD_OK;
}

// Forward conv of a conv+BN+SiLU unit: y_raw[M][ldy] (bf16) + BatchNorm partials stats[2][N][slots].
int kodhip_conv_fwd_raw(const void* x, const void* w_packed, void* y, float* stats,
                        int B, int H, int W, int ldx, int xcoff, int Cin,
                        int N, int KH, int KW, int SH, int SW, int PH, int PW, int Kp,
                        int ldy, int ycoff, hipStream_t stream) {
  ConvArgs a = {};
  // wide-pixel form (the stem: Cin = 32 = four 8-channel pixel pairs per tap, KW = 1): the window's last pixel is
  // K-alignment padding with zero weights, so the true kernel width for the output size is Cin/ldx - 1
  const int kw_out = (Cin > ldx && KW == 1) ? Cin / ldx - 1 : KW;
  int Ho = (H + 2 * PH - KH) / SH + 1, Wo = (W + 2 * PW - kw_out) / SW + 1;
  int rc = fill_common(a, x, w_packed, B, H, W, ldx, xcoff, Cin, Ho, Wo, N, KH, KW, Kp);
  if (rc) return rc;
  KOD_CHECK_ARG(y && stats, "conv_fwd_raw: null output");
  KOD_CHECK_ARG(N % 8 == 0 && ldy % 8 == 0 && ycoff % 8 == 0 && ycoff + N <= ldy, "conv_fwd_raw: bad output slice");
  a.y = (bf16_t*)y; a.stats = stats; a.ldy = ldy; a.ycoff = ycoff;
  a.stats_slots = kodhip_conv_stats_slots(a.M, N);
  a.mul_h = SH; a.mul_w = SW; a.add_h = -PH; a.add_w = -PW; a.tap_sign = 1; a.sh_shift = 0; a.sw_shift = 0;
  return launch<MODE_RAW>(a, stream);
}

// Fused detection head of one level: out[B][A][Ho*Wo][P] fp32 = 1x1 conv (N = A*(5+nc) packed as
// box(4A) | obj(A) | cls(nc*A)) + bias.
int kodhip_conv_fwd_head(const void* x, const void* w_packed, const float* bias, float* out,
                         int B, int H, int W, int ldx, int xcoff, int Cin, int A, int nc, int Kp,
                         hipStream_t stream) {
  ConvArgs a = {};
  int N = A * (5 + nc);
  int rc = fill_common(a, x, w_packed, B, H, W, ldx, xcoff, Cin, H, W, N, 1, 1, Kp);
  if (rc) return rc;
  KOD_CHECK_ARG(bias && out && A > 0 && nc > 0, "conv_fwd_head: bad args");
  a.bias = bias; a.head_out = out; a.head_A = A; a.head_P = 5 + nc; a.head_nc = nc; a.magic_hp = magic_u32((uint32_t)(5 + nc));
  KOD_CHECK_ARG(5 + nc <= 128, "conv_fwd_head: at most 123 classes");
  a.mul_h = 1; a.mul_w = 1; a.add_h = 0; a.add_w = 0; a.tap_sign = 1;
  return launch<MODE_HEAD>(a, stream);
}

// ---- data gradient ---------------------------------------------------------------------------------------------
}  // extern "C"

namespace {

struct BnRedSeg { int ch_begin, ch_count; const void* raw; int ldr; const float* aff; float* partials; };

int set_segments(ConvArgs& a, const BnRedSeg* segs, int nseg, int slots, int out_channels) {
  KOD_CHECK_ARG(segs && nseg >= 1 && nseg <= MAX_SEG && slots > 0, "conv_dgrad_bnred: 1..%d segments and a slot count expected", MAX_SEG);
  a.nseg = nseg; a.stats_slots = slots;
  for (int i = 0; i < nseg; ++i) {
    const BnRedSeg& g = segs[i];
    KOD_CHECK_ARG(g.raw && g.aff && g.partials && g.ch_count > 0 && g.ch_begin >= 0 && g.ch_begin % 8 == 0 && g.ch_count % 8 == 0 &&
                  g.ch_begin + g.ch_count <= out_channels && g.ldr % 8 == 0 && g.ldr >= g.ch_count, "conv_dgrad_bnred: bad segment %d", i);
    a.seg_begin[i] = g.ch_begin; a.seg_end[i] = g.ch_begin + g.ch_count; a.seg_ldr[i] = g.ldr; a.seg_C[i] = g.ch_count;
    a.seg_raw[i] = (const bf16_t*)g.raw; a.seg_aff[i] = g.aff; a.seg_part[i] = g.partials;
  }
  return KOD_OK;
}

// the `accumulate` argument of the data-gradient entry points: bit 0 = add to the bf16 partial already in dx (read-
// modify-write, rounds twice); bits 8.. = f32_mode (ConvArgs) with dx_f32 = the output buffer's fp32 shadow
int set_f32(ConvArgs& a, int accumulate, void* dx_f32) {
  a.accumulate = accumulate & 1;
  a.f32_mode = accumulate >> 8;
  a.y32 = (float*)dx_f32;
  KOD_CHECK_ARG(a.f32_mode >= 0 && a.f32_mode <= 4, "conv_dgrad: bad fp32 accumulation mode %d", a.f32_mode);
  KOD_CHECK_ARG(a.f32_mode == 0 || a.f32_mode == 4 || dx_f32, "conv_dgrad: fp32 accumulation mode %d needs the fp32 shadow of dx", a.f32_mode);
  return KOD_OK;
}

int prep_dgrad(ConvArgs& a, const void* dy, const void* w_dgrad, void* dx, int B, int H, int W, int ldx, int xcoff, int Cin,
               int N, int KH, int KW, int SH, int SW, int PH, int PW, int Kp, int ldy, int ycoff, int accumulate,
               void* dx_f32 = nullptr) {
  a = ConvArgs{};
  int Ho = (H + 2 * PH - KH) / SH + 1, Wo = (W + 2 * PW - KW) / SW + 1;
  // gather source is dy (channels N per tap), output pixels are the input pixels of the forward conv
  int rc = fill_common(a, dy, w_dgrad, B, Ho, Wo, ldy, ycoff, N, H, W, Cin, KH, KW, Kp);
  if (rc) return rc;
  KOD_CHECK_ARG(dx, "conv_dgrad: null output");
  KOD_CHECK_ARG(Cin % 8 == 0 && ldx % 8 == 0 && xcoff % 8 == 0 && xcoff + Cin <= ldx, "conv_dgrad: bad output slice");
  int ssh = ilog2_exact(SH), ssw = ilog2_exact(SW);
  KOD_CHECK_ARG(ssh >= 0 && ssw >= 0, "conv_dgrad: stride must be a power of two");
  a.y = (bf16_t*)dx; a.ldy = ldx; a.ycoff = xcoff;
  if (int rc2 = set_f32(a, accumulate, dx_f32)) return rc2;
  a.mul_h = 1; a.mul_w = 1; a.add_h = PH; a.add_w = PW; a.tap_sign = -1; a.sh_shift = ssh; a.sw_shift = ssw;
  return KOD_OK;
}

// the four parity classes of a 3x3 / stride 2 / pad 1 data gradient (see kodhip_conv_dgrad_s2)
int prep_dgrad_s2(ConvArgs cls[4], bool& all_fast, const void* dy, const void* w_dgrad_s2, void* dx, int B, int H, int W,
                  int ldx, int xcoff, int Cin, int N, int ldy, int ycoff, int accumulate, void* dx_f32 = nullptr) {
  KOD_CHECK_ARG(H % 2 == 0 && W % 2 == 0, "conv_dgrad_s2: input dims must be even");
  const int Ho = H / 2, Wo = W / 2;
  size_t woff = 0;
  all_fast = !getenv("KODHIP_S2_SEPARATE");
  for (int c = 0; c < 4; ++c) {
    const int py = c >> 1, px = c & 1;
    const int KH = 1 + py, KW = 1 + px;
    const int Kp = KH * KW * ((N + 31) / 32 * 32);
    ConvArgs& a = cls[c];
    a = ConvArgs{};
    // gather source dy [B,Ho,Wo,N]; class outputs form a Ho x Wo grid scattered into dx with stride 2
    int rc = fill_common(a, dy, (const bf16_t*)w_dgrad_s2 + woff, B, Ho, Wo, ldy, ycoff, N, Ho, Wo, Cin, KH, KW, Kp);
    if (rc) return rc;
    KOD_CHECK_ARG(dx && Cin % 8 == 0 && ldx % 8 == 0 && xcoff % 8 == 0 && xcoff + Cin <= ldx, "conv_dgrad_s2: bad output slice");
    a.y = (bf16_t*)dx; a.ldy = ldx; a.ycoff = xcoff;
    if (int rc2 = set_f32(a, accumulate, dx_f32)) return rc2;
    a.mul_h = 1; a.mul_w = 1; a.add_h = py; a.add_w = px; a.tap_sign = -1; a.sh_shift = 0; a.sw_shift = 0;
    a.out_mul = 2; a.out_off_y = py; a.out_off_x = px; a.out_H = H; a.out_W = W;
    all_fast = all_fast && fast_eligible(a);
    woff += (size_t)Cin * Kp;
  }
  return KOD_OK;
}

// the same data gradient "folded": ONE stride-1 gather over the 2x2 dY neighbourhood {i, i+1} x {j, j+1} of the output
// pixel block (2i..2i+1, 2j..2j+1) with N = 4 classes x Cin columns (class (py,px) uses tap (dy,dx) iff py >= dy and
// px >= dx; the other weights are zero) and a depth-to-space epilogue.  16 tap-class products instead of 9, but dY is
// staged once instead of 9/4 times, the K loop is 4x round_up(N,32) long for every tile, and the epilogue writes
// both x parities of a pixel pair: the shallow layers (Cin <= 64), which are bound by staging / pipeline fill and
// half-line stores and not by the MFMA, run 2x faster this way.
int prep_dgrad_s2f(ConvArgs& a, const void* dy, const void* w_fold, void* dx, int B, int H, int W,
                   int ldx, int xcoff, int Cin, int N, int ldy, int ycoff, int accumulate, void* dx_f32 = nullptr) {
  KOD_CHECK_ARG(H % 2 == 0 && W % 2 == 0, "conv_dgrad_s2f: input dims must be even");
  const int Ho = H / 2, Wo = W / 2;
  const int Kp = 4 * ((N + 31) / 32 * 32);
  a = ConvArgs{};
  if (int rc = fill_common(a, dy, w_fold, B, Ho, Wo, ldy, ycoff, N, Ho, Wo, 4 * Cin, 2, 2, Kp)) return rc;
  KOD_CHECK_ARG(dx && Cin % 8 == 0 && ldx % 8 == 0 && xcoff % 8 == 0 && xcoff + Cin <= ldx, "conv_dgrad_s2f: bad output slice");
  a.y = (bf16_t*)dx; a.ldy = ldx; a.ycoff = xcoff;
  if (int rc2 = set_f32(a, accumulate, dx_f32)) return rc2;
  a.mul_h = 1; a.mul_w = 1; a.add_h = 0; a.add_w = 0; a.tap_sign = 1; a.sh_shift = 0; a.sw_shift = 0;
  a.out_mul = 2; a.out_off_y = 0; a.out_off_x = 0; a.out_H = H; a.out_W = W; a.d2s_C = Cin;
  static const bool skip = !(getenv("KODHIP_S2F_SKIP") && getenv("KODHIP_S2F_SKIP")[0] == '0');      // A/B knob
  a.d2s_skip = skip ? 1 : 0;
  KOD_CHECK_ARG(fast_eligible(a), "conv_dgrad_s2f: needs the LDS-DMA path (operands within a 32-bit buffer range)");
  return KOD_OK;
}

}  // namespace

extern "C" {

// Data gradient: dx[B][H][W][ldx](+xcoff, Cin channels) (+)= conv_transpose(dy[B][Ho][Wo][ldy](+ycoff, N), w).
// w_dgrad is packed [Cin][Kp] with k = (kh, kw, n).
int kodhip_conv_dgrad(const void* dy, const void* w_dgrad, void* dx,
                      int B, int H, int W, int ldx, int xcoff, int Cin,
                      int N, int KH, int KW, int SH, int SW, int PH, int PW, int Kp,
                      int ldy, int ycoff, int accumulate, void* dx_f32, hipStream_t stream) {
  ConvArgs a;
  if (int rc = prep_dgrad(a, dy, w_dgrad, dx, B, H, W, ldx, xcoff, Cin, N, KH, KW, SH, SW, PH, PW, Kp, ldy, ycoff, accumulate, dx_f32)) return rc;
  return launch<MODE_PLAIN>(a, stream);
}

// Data gradient of a 3x3 / stride 2 / pad 1 convolution, decomposed by output-pixel parity: class (py,px)
// only meets taps kh = 1 (py=0) or kh in {0,2} (py=1) (same for kw), so the four classes are stride-1 gathers
// with 1, 2, 2 and 4 taps - 9 taps of MFMA work instead of 36.  w_dgrad_s2 holds the four class packs
// back to back: class c = 2*py+px is [Cin][Kdp_c], Kdp_c = ntaps_c * round_up(N, 32), k = (kh', kw', n) tap-major.
int kodhip_conv_dgrad_s2(const void* dy, const void* w_dgrad_s2, void* dx,
                         int B, int H, int W, int ldx, int xcoff, int Cin, int N,
                         int ldy, int ycoff, int accumulate, void* dx_f32, hipStream_t stream) {
  ConvArgs cls[4];
  bool all_fast;
  if (int rc = prep_dgrad_s2(cls, all_fast, dy, w_dgrad_s2, dx, B, H, W, ldx, xcoff, Cin, N, ldy, ycoff, accumulate, dx_f32)) return rc;
  if (all_fast) return launch_x4<MODE_PLAIN>(cls, stream);
  for (int c = 0; c < 4; ++c)
    if (int rc = launch<MODE_PLAIN>(cls[c], stream)) return rc;
  return KOD_OK;
}

// Which form the engine should pack and call for a 3x3 / stride 2 / pad 1 layer with Cin input and N output channels:
// 1 = folded (kodhip_conv_dgrad_s2f, pack mode 3), 0 = parity classes (kodhip_conv_dgrad_s2, pack mode 2).
// KODHIP_S2_FOLD_MAXC overrides the channel threshold (0 = never fold).
int kodhip_conv_dgrad_s2_folded(int Cin, int N) {
  static int maxc = -1;
  if (maxc < 0) { const char* e = getenv("KODHIP_S2_FOLD_MAXC"); maxc = e ? atoi(e) : 64; }
  (void)N;
  return Cin <= maxc ? 1 : 0;
}

// Folded form of kodhip_conv_dgrad_s2 (see prep_dgrad_s2f).  w_fold: [4 * Cin][4 * round_up(N, 32)], row = class * Cin
// + ci (class = 2 * py + px), k = (dy * 2 + dx) * round_up(N, 32) + n.
int kodhip_conv_dgrad_s2f(const void* dy, const void* w_fold, void* dx,
                          int B, int H, int W, int ldx, int xcoff, int Cin, int N,
                          int ldy, int ycoff, int accumulate, void* dx_f32, hipStream_t stream) {
  ConvArgs a;
  if (int rc = prep_dgrad_s2f(a, dy, w_fold, dx, B, H, W, ldx, xcoff, Cin, N, ldy, ycoff, accumulate, dx_f32)) return rc;
  return launch<MODE_PLAIN>(a, stream);
}

int kodhip_conv_dgrad_s2f_bnred_slots(int B, int H, int W, int Cin, int N, int ldy) {
  const void* fake = (const void*)64;
  if (getenv("KODHIP_NO_BNRED")) return 0;
  ConvArgs a;
  if (prep_dgrad_s2f(a, fake, fake, (void*)fake, B, H, W, Cin, 0, Cin, N, ldy, 0, 0)) return 0;
  return 4 * make_plan(a.M, a.N, a.K, true, false, fold_bn_cap(a)).groups_m;
}

int kodhip_conv_dgrad_s2f_bnred(const void* dy, const void* w_fold, void* dx,
                                int B, int H, int W, int ldx, int xcoff, int Cin, int N,
                                int ldy, int ycoff, int accumulate, void* dx_f32, const void* segments, int nseg, int slots,
                                hipStream_t stream) {
  ConvArgs a;
  if (int rc = prep_dgrad_s2f(a, dy, w_fold, dx, B, H, W, ldx, xcoff, Cin, N, ldy, ycoff, accumulate, dx_f32)) return rc;
  if (int rc = set_segments(a, (const BnRedSeg*)segments, nseg, slots, Cin)) return rc;
  return launch<MODE_PLAIN_BN>(a, stream);
}

// ---- data gradient + BatchNorm-backward reduction of the units whose output gradient this launch completes.
// The launch must be the LAST writer of dx's channel ranges named by the segments; per segment it leaves
// partials[2][ch_count][slots] = per-block sums of dz and dz*y (dz = dx * silu'(y*scale + shift)), which
// kodhip_bn_bwd_coeffs_partials(..., raw_moment = 1) turns into the BatchNorm-backward coefficients.
// *_slots: slots a launch of this geometry writes (allocate partials with exactly that many); 0 = this geometry
// cannot carry the fused reduction (run kodhip_bn_silu_bwd_reduce instead).  stride2 = the 3x3/s2/p1 form.
int kodhip_conv_dgrad_bnred_slots(int B, int H, int W, int Cin, int N, int KH, int KW, int SH, int SW, int PH, int PW,
                                  int ldy, int stride2) {
  const void* fake = (const void*)64;      // never dereferenced: only the launch plan is computed
  if (getenv("KODHIP_NO_BNRED")) return 0;
  if (stride2) {
    ConvArgs cls[4];
    bool all_fast;
    if (prep_dgrad_s2(cls, all_fast, fake, fake, (void*)fake, B, H, W, Cin, 0, Cin, N, ldy, 0, 0) || !all_fast) return 0;
    int kmax = 0;
    for (int i = 0; i < 4; ++i) kmax = cls[i].K > kmax ? cls[i].K : kmax;
    return 4 * make_plan(cls[0].M, cls[0].N, kmax, true).groups_m;
  }
  ConvArgs a;
  const int Kp = KH * KW * ((N + 31) / 32 * 32);
  if (prep_dgrad(a, fake, fake, (void*)fake, B, H, W, Cin, 0, Cin, N, KH, KW, SH, SW, PH, PW, Kp, ldy, 0, 0) || !fast_eligible(a)) return 0;
  bool row3;
  return plan_conv(a, true, row3, MODE_PLAIN_BN).groups_m;
}

int kodhip_conv_dgrad_bnred(const void* dy, const void* w_dgrad, void* dx,
                            int B, int H, int W, int ldx, int xcoff, int Cin,
                            int N, int KH, int KW, int SH, int SW, int PH, int PW, int Kp,
                            int ldy, int ycoff, int accumulate, void* dx_f32, const void* segments, int nseg, int slots,
                            hipStream_t stream) {
  ConvArgs a;
  if (int rc = prep_dgrad(a, dy, w_dgrad, dx, B, H, W, ldx, xcoff, Cin, N, KH, KW, SH, SW, PH, PW, Kp, ldy, ycoff, accumulate, dx_f32)) return rc;
  if (int rc = set_segments(a, (const BnRedSeg*)segments, nseg, slots, Cin)) return rc;
  return launch<MODE_PLAIN_BN>(a, stream);
}

// ---- data gradient of TWO pointwise convs that read the same tensor (a CSP layer's main_conv and short_conv,
// kod/nn/layers/csp.py:96-111): dx = conv_transpose(dy1, w1) + conv_transpose(dy2, w2) as one launch over the
// concatenated reduction [dy1 channels | dy2 channels] - dx is written once instead of written and then
// read-modified-written by a second launch.  1x1 / stride 1 / no padding; dy1, dy2: [B*H*W][ldy] (+ycoff, N channels
// each); w1, w2: [Cin][Kp] dgrad packs (Kp = round_up(N, 32)).  *_bnred: also the BatchNorm-backward reduction of the
// units whose output gradient dx completes (see kodhip_conv_dgrad_bnred).
static int prep_dgrad_dual(ConvArgs& a, const void* dy1, const void* w1, const void* dy2, const void* w2, void* dx,
                           int B, int H, int W, int ldx, int xcoff, int Cin, int N, int Kp, int ldy, int ycoff, int accumulate,
                           void* dx_f32 = nullptr) {
  if (int rc = prep_dgrad(a, dy1, w1, dx, B, H, W, ldx, xcoff, Cin, N, 1, 1, 1, 1, 0, 0, Kp, ldy, ycoff, accumulate, dx_f32)) return rc;
  KOD_CHECK_ARG(dy2 && w2, "conv_dgrad_dual: null second source");
  KOD_CHECK_ARG(fast_eligible(a), "conv_dgrad_dual: needs the LDS-DMA path (operands within a 32-bit buffer range)");
  a.x2 = (const bf16_t*)dy2; a.w2 = (const bf16_t*)w2; a.nk1 = Kp / 32;
  return KOD_OK;
}

int kodhip_conv_dgrad_dual(const void* dy1, const void* w1, const void* dy2, const void* w2, void* dx,
                           int B, int H, int W, int ldx, int xcoff, int Cin, int N, int Kp, int ldy, int ycoff,
                           int accumulate, void* dx_f32, hipStream_t stream) {
  ConvArgs a;
  if (int rc = prep_dgrad_dual(a, dy1, w1, dy2, w2, dx, B, H, W, ldx, xcoff, Cin, N, Kp, ldy, ycoff, accumulate, dx_f32)) return rc;
  return launch<MODE_PLAIN>(a, stream);
}

int kodhip_conv_dgrad_dual_bnred_slots(int B, int H, int W, int Cin, int N, int ldy) {
  const void* fake = (const void*)64;
  if (getenv("KODHIP_NO_BNRED")) return 0;
  ConvArgs a;
  const int Kp = (N + 31) / 32 * 32;
  if (prep_dgrad_dual(a, fake, fake, fake, fake, (void*)fake, B, H, W, Cin, 0, Cin, N, Kp, ldy, 0, 0)) return 0;
  return make_plan(a.M, a.N, 2 * a.K, true).groups_m;
}

int kodhip_conv_dgrad_dual_bnred(const void* dy1, const void* w1, const void* dy2, const void* w2, void* dx,
                                 int B, int H, int W, int ldx, int xcoff, int Cin, int N, int Kp, int ldy, int ycoff,
                                 int accumulate, void* dx_f32, const void* segments, int nseg, int slots, hipStream_t stream) {
  ConvArgs a;
  if (int rc = prep_dgrad_dual(a, dy1, w1, dy2, w2, dx, B, H, W, ldx, xcoff, Cin, N, Kp, ldy, ycoff, accumulate, dx_f32)) return rc;
  if (int rc = set_segments(a, (const BnRedSeg*)segments, nseg, slots, Cin)) return rc;
  return launch<MODE_PLAIN_BN>(a, stream);
}

int kodhip_conv_dgrad_s2_bnred(const void* dy, const void* w_dgrad_s2, void* dx,
                               int B, int H, int W, int ldx, int xcoff, int Cin, int N,
                               int ldy, int ycoff, int accumulate, void* dx_f32, const void* segments, int nseg, int slots,
                               hipStream_t stream) {
  ConvArgs cls[4];
  bool all_fast;
  if (int rc = prep_dgrad_s2(cls, all_fast, dy, w_dgrad_s2, dx, B, H, W, ldx, xcoff, Cin, N, ldy, ycoff, accumulate, dx_f32)) return rc;
  KOD_CHECK_ARG(all_fast, "conv_dgrad_s2_bnred: this geometry cannot carry the fused reduction (query the slots first)");
  for (int c = 0; c < 4; ++c)
    if (int rc = set_segments(cls[c], (const BnRedSeg*)segments, nseg, slots, Cin)) return rc;
  return launch_x4<MODE_PLAIN_BN>(cls, stream);
}

}  // extern "C"
