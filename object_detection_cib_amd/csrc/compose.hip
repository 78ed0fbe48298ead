// Device data path: mosaic paste + affine warp + HSV jitter + flip + /255 (+ mixup) in ONE gather kernel.
//
// Reference (per sample, in DataLoader workers, numpy / OpenCV):
//   mosaic   kod/data/mosaic.py:58-132          4 u8 HWC images pasted on a 2S x 2S canvas filled with 114
//   affine   kod/data/augmentations/default.py:279-320   cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT 114) -> S x S
//            (or cv2.warpPerspective when a perspective draw is non-zero, default.py:306-313)
//   colour   default.py:420-432,460-461         image_color_transforms: albumentations Blur / MedianBlur / ToGray / CLAHE, p = 0.01
//            each, on the warped u8 image - RARE PATH: a sample whose gate fired (~4 %) is warped into a u8 scratch image
//            by warp_u8_kernel, the fired transforms run on it (kernels at the end of this file), and the compositing
//            kernel then reads that sample's pixels from the scratch image instead of warping them (SampleDesc::pre)
//   hsv      default.py:354-383                 cvtColor(BGR2HSV) -> 3 LUTs -> cvtColor(HSV2BGR) (on RGB data: kept)
//   flip     default.py:386-397                 np.fliplr
//   tensor   default.py:433-438,482             ToFloat(255) + HWC->CHW
//   mixup    default.py:400-408                 im1*r + im2*(1-r)
//
// Here the canvas is never materialised: every output pixel inverse-maps into canvas coordinates with
// OpenCV's fixed-point arithmetic (AB_BITS=10, INTER_BITS=5, 15-bit bilinear weights), resolves each of the
// four taps to a source-image pixel (or 114) through the mosaic rectangles, and runs the integer HSV round
// trip in registers.  Source images live in one u8 pool in HBM (the RAM cache of kod/data/detection.py:66-76).
// Host code supplies, per sample, the draws in the reference's RNG order (see data/device_pipeline.py).
#include "kodhip_common.h"

namespace {

struct TileDesc {            // one mosaic tile
  long off;                  // byte offset of the source image in the pool (HWC u8)
  int h, w;                  // source image size
  int x1a, y1a, x2a, y2a;    // destination rectangle on the canvas
  int x1b, y1b;              // source rectangle origin
};
struct SampleDesc {
  TileDesc tile[4];
  double im[6];              // inverse affine matrix (row major 2x3), as OpenCV computes it; persp: rows 0 - 1 of the inverse 3x3
  double pw[3];              // persp: row 2 of the inverse 3x3 matrix
  unsigned char lut_h[256], lut_s[256], lut_v[256];
  int hsv_on;
  int flip;
  int canvas;                // 2S
  int persp;                 // 1: cv2.warpPerspective's mapping (default.py:306-313) instead of cv2.warpAffine's
  // image_color_transforms: what the colour stage's gate drew for this sample (bit 0 Blur, 1 MedianBlur, 2 ToGray, 3 CLAHE; host
  // side: data/host_protocol.color_gate) and, once kodhip_compose_color has run, the address of the S x S x 3 u8 image it left
  int color, blur_k, median_k, pad0;
  const unsigned char* pre;  // nullptr: warp from the pool
  double clahe_clip;
};

__constant__ short c_tab_dummy;   // (keeps the TU non-empty for some toolchains)

// byte offset in the pool of canvas pixel (y, x), or -1 where the canvas shows its fill value 114
__device__ __forceinline__ long locate(const SampleDesc& d, int y, int x) {
  if (x < 0 || y < 0 || x >= d.canvas || y >= d.canvas) return -1;
  // quadrant by destination rectangles (tiles never overlap)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const TileDesc& t = d.tile[i];
    if (x >= t.x1a && x < t.x2a && y >= t.y1a && y < t.y2a)
      return t.off + ((long)(y - t.y1a + t.y1b) * t.w + (x - t.x1a + t.x1b)) * 3;
  }
  return -1;
}

// OpenCV RGB2HSV_b (hsv_shift = 12, hrange = 180) on (b,g,r) = the three stored channels in order.  sdiv / hdiv: OpenCV's
// own tables (sdiv_table[v] = round((255 << 12) / v), hdiv_table180[diff] = round((180 << 12) / (6 diff)), entry 0 = 0),
// computed once on the host in the same fp64 arithmetic - two fp64 divisions per pixel otherwise
__device__ __forceinline__ void bgr2hsv(int b, int g, int r, const int* sdiv_tab, const int* hdiv_tab, int& h, int& s, int& v) {
  v = max(max(b, g), r);
  int vmin = min(min(b, g), r);
  int diff = v - vmin;
  int vr = (v == r) ? -1 : 0, vg = (v == g) ? -1 : 0;
  int sdiv = sdiv_tab[v];
  int hdiv = hdiv_tab[diff];
  s = (diff * sdiv + (1 << 11)) >> 12;
  int hh = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
  hh = (hh * hdiv + (1 << 11)) >> 12;
  hh += hh < 0 ? 180 : 0;
  h = hh;
}

// OpenCV HSV2RGB_b for u8: float path, h*(6/180), s/255, v/255, result saturate_cast<uchar>(x*255)
__device__ __forceinline__ void hsv2bgr(int h, int s, int v, int& b, int& g, int& r) {
  float fh = (float)h * (6.f / 180.f), fs = (float)s * (1.f / 255.f), fv = (float)v * (1.f / 255.f);
  float ob, og, orr;
  if (fs == 0.f) {
    ob = og = orr = fv;
  } else {
    if (fh < 0.f) fh += 6.f; else if (fh >= 6.f) fh -= 6.f;
    int sector = (int)floorf(fh);
    fh -= (float)sector;
    if ((unsigned)sector >= 6u) { sector = 0; fh = 0.f; }
    const float t0 = fv, t1 = fv * (1.f - fs), t2 = fv * (1.f - fs * fh), t3 = fv * (1.f - fs * (1.f - fh));
    // sector table {1,3,0},{1,0,2},{3,0,1},{0,2,1},{0,1,3},{2,1,0} as selects (no indexed local array: scratch memory)
    ob = sector == 0 ? t1 : sector == 1 ? t1 : sector == 2 ? t3 : sector == 3 ? t0 : sector == 4 ? t0 : t2;
    og = sector == 0 ? t3 : sector == 1 ? t0 : sector == 2 ? t0 : sector == 3 ? t2 : sector == 4 ? t1 : t1;
    orr = sector == 0 ? t0 : sector == 1 ? t2 : sector == 2 ? t1 : sector == 3 ? t1 : sector == 4 ? t3 : t0;
  }
  b = min(max((int)rintf(ob * 255.f), 0), 255);
  g = min(max((int)rintf(og * 255.f), 0), 255);
  r = min(max((int)rintf(orr * 255.f), 0), 255);
}

// the warped (pre-HSV, unflipped) pixel (y, x) of sample d: cv2.warpAffine / warpPerspective over the never-materialised canvas
__device__ __forceinline__ void warp_px(const unsigned char* pool, const SampleDesc& d, const short* tab, int y, int x, int out[3]) {
  // cv::warpAffine: X0 = round((M01*y + M02)*1024) + 16, adelta = round(M00*x*1024); coords in 1/32 px
  int X, Y;
  if (d.persp) {
    // cv::warpPerspective: X0 = M00 x + M01 y + M02, W = M20 x + M21 y + M22 in doubles, W = W ? 32 / W : 0,
    // X = saturate_cast<int>(clamp(X0 W)) (half to even), source pixel X >> 5 saturated to int16, weights index X & 31
    const double dx = (double)x, dy = (double)y;
    const double X0 = (d.im[0] * dx + d.im[1] * dy) + d.im[2];
    const double Y0 = (d.im[3] * dx + d.im[4] * dy) + d.im[5];
    double W = (d.pw[0] * dx + d.pw[1] * dy) + d.pw[2];
    W = W != 0.0 ? 32.0 / W : 0.0;
    const double fX = fmax(-2147483648.0, fmin(2147483647.0, X0 * W)), fY = fmax(-2147483648.0, fmin(2147483647.0, Y0 * W));
    X = (int)__double2ll_rn(fX); Y = (int)__double2ll_rn(fY);
  } else {
    const long AB = 1024;
    long X0 = __double2ll_rn((d.im[1] * y + d.im[2]) * (double)AB) + 16;
    long Y0 = __double2ll_rn((d.im[4] * y + d.im[5]) * (double)AB) + 16;
    long ad = __double2ll_rn(d.im[0] * x * (double)AB);
    long bd = __double2ll_rn(d.im[3] * x * (double)AB);
    X = (int)((X0 + ad) >> 5); Y = (int)((Y0 + bd) >> 5);
  }
  int sx = X >> 5, sy = Y >> 5, fx = X & 31, fy = Y & 31;
  if (d.persp) { sx = min(max(sx, -32768), 32767); sy = min(max(sy, -32768), 32767); }
  const short* w = tab + (fy * 32 + fx) * 4;
  // each of the four taps is resolved to its source pixel ONCE (not once per channel), and its three channel bytes come
  // with ONE unaligned 4-byte load (the pool carries 4 bytes of slack behind the last image: kodhip_compose_batch's contract)
  const long o00 = locate(d, sy, sx), o01 = locate(d, sy, sx + 1), o10 = locate(d, sy + 1, sx), o11 = locate(d, sy + 1, sx + 1);
  const int w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
  const unsigned fill = 114u | (114u << 8) | (114u << 16);
  unsigned q00 = fill, q01 = fill, q10 = fill, q11 = fill;
  if (o00 >= 0) __builtin_memcpy(&q00, pool + o00, 4);
  if (o01 >= 0) __builtin_memcpy(&q01, pool + o01, 4);
  if (o10 >= 0) __builtin_memcpy(&q10, pool + o10, 4);
  if (o11 >= 0) __builtin_memcpy(&q11, pool + o11, 4);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int p00 = (q00 >> (8 * c)) & 255, p01 = (q01 >> (8 * c)) & 255, p10 = (q10 >> (8 * c)) & 255, p11 = (q11 >> (8 * c)) & 255;
    out[c] = (p00 * w0 + p01 * w1 + p10 * w2 + p11 * w3 + (1 << 14)) >> 15;
  }
}

// one composite pixel (3 channels, u8 domain) of sample d at output (y, xo)
__device__ __forceinline__ void composite(const unsigned char* pool, const SampleDesc& d, const short* tab,
                                          const int* sdiv_tab, const int* hdiv_tab, int y, int xo, int S, int out[3]) {
  const int x = d.flip ? S - 1 - xo : xo;
  if (d.pre) {               // the colour stage ran on this sample: its warped pixels are in the scratch image
    const unsigned char* q = d.pre + ((size_t)y * S + x) * 3;
    out[0] = q[0]; out[1] = q[1]; out[2] = q[2];
  } else {
    warp_px(pool, d, tab, y, x, out);
  }
  if (d.hsv_on) {
    int h, s, v, b, g, r;
    bgr2hsv(out[0], out[1], out[2], sdiv_tab, hdiv_tab, h, s, v);
    hsv2bgr(d.lut_h[h], d.lut_s[s], d.lut_v[v], b, g, r);
    out[0] = b; out[1] = g; out[2] = r;
  }
}

// grid: (ceil(S*S/2/256), B): one thread per horizontal pixel PAIR (one 16-byte store of the network's input layout).
// descs: [B][2] SampleDesc (second = mixup partner), mix[b] < 0 => no mixup.  tab: 32*32*4 int16 bilinear weights followed by
// sdiv_table[256] | hdiv_table180[256] (int32).
__global__ __launch_bounds__(256) void compose_kernel(const unsigned char* pool, const SampleDesc* descs,
                                                      const float* mix, const short* tab, float* out_f32,
                                                      bf16_t* out_pairs, int S) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int half = S >> 1;
  if (p >= S * half) return;
  const int y = p / half, x0 = (p - y * half) * 2;
  const int* sdiv_tab = reinterpret_cast<const int*>(tab + 32 * 32 * 4);
  const int* hdiv_tab = sdiv_tab + 256;
  const float r = mix[b * 2], r1 = mix[b * 2 + 1];
  float v[2][3];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    int px[3];
    composite(pool, descs[b * 2], tab, sdiv_tab, hdiv_tab, y, x0 + k, S, px);
#pragma unroll
    for (int c = 0; c < 3; ++c) v[k][c] = (float)px[c] / 255.f;
    if (r >= 0.f) {
      int q[3];
      composite(pool, descs[b * 2 + 1], tab, sdiv_tab, hdiv_tab, y, x0 + k, S, q);
#pragma unroll
      for (int c = 0; c < 3; ++c) v[k][c] = v[k][c] * r + ((float)q[c] / 255.f) * r1;
    }
  }
  if (out_f32) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float2 o = {v[0][c], v[1][c]};
      *reinterpret_cast<float2*>(out_f32 + ((size_t)(b * 3 + c) * S + y) * S + x0) = o;
    }
  }
  if (out_pairs) {   // network input layout [B][S][S/2][8] = pixel pairs x 4 channels
    bf16x8 o = {(bf16_t)v[0][0], (bf16_t)v[0][1], (bf16_t)v[0][2], (bf16_t)0.f, (bf16_t)v[1][0], (bf16_t)v[1][1], (bf16_t)v[1][2], (bf16_t)0.f};
    *reinterpret_cast<bf16x8*>(out_pairs + ((size_t)(b * S + y) * S + x0) * 4) = o;
  }
}

// ------------------------------------------------------------------ image_color_transforms (rare path, see the header)
// All four follow oracle/datapath.py's restatements operation for operation (integer arithmetic and tables; CLAHE's
// interpolation in fp32 without contraction), so the product and the oracle agree bit for bit; OpenCV's / albumentations'
// own arithmetic is not available in this environment (parity unpinned).
__global__ __launch_bounds__(256) void warp_u8_kernel(const unsigned char* pool, const SampleDesc* descs, const short* tab,
                                                      int entry, unsigned char* out, int S) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= S * S) return;
  int px[3];
  warp_px(pool, descs[entry], tab, p / S, p % S, px);
  out[(size_t)p * 3 + 0] = (unsigned char)px[0]; out[(size_t)p * 3 + 1] = (unsigned char)px[1]; out[(size_t)p * 3 + 2] = (unsigned char)px[2];
}

__device__ __forceinline__ int reflect101(int i, int n) { i = i < 0 ? -i : i; return i >= n ? 2 * (n - 1) - i : i; }

// cv2.blur(img, (k, k)): box filter, BORDER_REFLECT_101, cvRound(sum * (1 / k^2))
__global__ __launch_bounds__(256) void blur_u8_kernel(const unsigned char* src, unsigned char* dst, int S, int k) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= S * S) return;
  const int y = p / S, x = p % S, r = k >> 1;
  int s0 = 0, s1 = 0, s2 = 0;
  for (int dy = -r; dy <= r; ++dy) {
    const unsigned char* row = src + (size_t)reflect101(y + dy, S) * S * 3;
    for (int dx = -r; dx <= r; ++dx) {
      const unsigned char* q = row + reflect101(x + dx, S) * 3;
      s0 += q[0]; s1 += q[1]; s2 += q[2];
    }
  }
  const double sc = 1.0 / (double)(k * k);
  dst[(size_t)p * 3 + 0] = (unsigned char)min(max(__double2int_rn((double)s0 * sc), 0), 255);
  dst[(size_t)p * 3 + 1] = (unsigned char)min(max(__double2int_rn((double)s1 * sc), 0), 255);
  dst[(size_t)p * 3 + 2] = (unsigned char)min(max(__double2int_rn((double)s2 * sc), 0), 255);
}

// cv2.medianBlur(img, K): per-channel median of the K x K window, BORDER_REPLICATE.  The median of n = K^2 bytes is the
// largest t with #{v >= t} > n / 2: found bit by bit (8 counting passes over the window held in registers).
template <int K>
__global__ __launch_bounds__(256) void median_u8_kernel(const unsigned char* src, unsigned char* dst, int S) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= S * S) return;
  const int y = p / S, x = p % S, r = K / 2;
#pragma unroll 1
  for (int c = 0; c < 3; ++c) {
    int v[K * K];
#pragma unroll
    for (int dy = 0; dy < K; ++dy)
#pragma unroll
      for (int dx = 0; dx < K; ++dx)
        v[dy * K + dx] = src[((size_t)min(max(y + dy - r, 0), S - 1) * S + min(max(x + dx - r, 0), S - 1)) * 3 + c];
    int med = 0;
#pragma unroll 1
    for (int bit = 7; bit >= 0; --bit) {
      const int cand = med | (1 << bit);
      int cnt = 0;
#pragma unroll
      for (int i = 0; i < K * K; ++i) cnt += v[i] >= cand ? 1 : 0;
      if (cnt > (K * K) / 2) med = cand;
    }
    dst[(size_t)p * 3 + c] = (unsigned char)med;
  }
}

// albumentations ToGray: RGB2GRAY (15-bit weights on channels 0, 1, 2) replicated to three channels; in place
__global__ __launch_bounds__(256) void gray_u8_kernel(unsigned char* img, int S) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= S * S) return;
  unsigned char* q = img + (size_t)p * 3;
  const int yv = (q[0] * 9798 + q[1] * 19235 + q[2] * 3735 + (1 << 14)) >> 15;
  q[0] = q[1] = q[2] = (unsigned char)yv;
}

// tables of the 8-bit RGB <-> Lab round trip (data/device_pipeline.color_table(); oracle/datapath.lab_tables): byte offsets
struct LabTab {
  const unsigned short* gamma;   // [256]
  const unsigned short* cbrt;    // [3072]
  const int* C;                  // [9] forward matrix, Q12
  const int* Cinv;               // [9] inverse matrix, Q12
  const int* fy; const int* dfx; const int* dfz;     // [256] each, Q15
  const unsigned char* enc;      // [4096]
};
__device__ __forceinline__ LabTab lab_tab(const unsigned char* t) {
  LabTab L;
  L.gamma = (const unsigned short*)t; L.cbrt = (const unsigned short*)(t + 512);
  L.C = (const int*)(t + 6656); L.Cinv = (const int*)(t + 6692);
  L.fy = (const int*)(t + 6728); L.dfx = (const int*)(t + 7752); L.dfz = (const int*)(t + 8776);
  L.enc = t + 9800;
  return L;
}
__device__ __forceinline__ void rgb2lab(const LabTab& T, int r, int g, int b, int& L, int& A, int& B) {
  const long R = T.gamma[r], G = T.gamma[g], Bl = T.gamma[b];
  long f[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) f[i] = T.cbrt[(R * T.C[3 * i] + G * T.C[3 * i + 1] + Bl * T.C[3 * i + 2] + (1 << 11)) >> 12];
  const long Lscale = (116 * 255 + 50) / 100, Lshift = -((16l * 255 * (1 << 15) + 50) / 100);
  L = (int)min(max((Lscale * f[1] + Lshift + (1 << 14)) >> 15, 0l), 255l);
  A = (int)min(max((500 * (f[0] - f[1]) + 128l * (1 << 15) + (1 << 14)) >> 15, 0l), 255l);
  B = (int)min(max((200 * (f[1] - f[2]) + 128l * (1 << 15) + (1 << 14)) >> 15, 0l), 255l);
}
__device__ __forceinline__ void lab2rgb(const LabTab& T, int L, int A, int B, int out[3]) {
  const long y = T.fy[L];
  long f[3] = {y + T.dfx[A], y, y - T.dfz[B]}, xyz[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const long t = min(max(f[i], 0l), 65535l);
    xyz[i] = t > 6780 ? (t * t * t + (1l << 28)) >> 29 : max((16832l * (t - 4520) * 16 + (1l << 19)) >> 20, 0l);      // Q16
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const long lin = (xyz[0] * T.Cinv[3 * i] + xyz[1] * T.Cinv[3 * i + 1] + xyz[2] * T.Cinv[3 * i + 2] + (1 << 15)) >> 16;
    out[i] = T.enc[min(max(lin, 0l), 4095l)];
  }
}

// cv::CLAHE (8 x 8 tiles) on the L channel: one block per tile builds the clipped, redistributed histogram of the (reflect-
// padded) tile and its cumulative look-up table luts[tile][256]
__global__ __launch_bounds__(256) void clahe_hist_kernel(const unsigned char* src, unsigned char* luts, const unsigned char* ctab,
                                                         int S, int cl) {
  __shared__ int hist[256];
  __shared__ int red[256];
  const LabTab T = lab_tab(ctab);
  const int t = threadIdx.x, ty = blockIdx.x >> 3, tx = blockIdx.x & 7;
  const int ts = (S + 7) >> 3;                    // tile side on the image padded to a multiple of 8
  hist[t] = 0;
  __syncthreads();
  for (int i = t; i < ts * ts; i += 256) {
    const int py = ty * ts + i / ts, px = tx * ts + i % ts;
    const unsigned char* q = src + ((size_t)reflect101(py, S) * S + reflect101(px, S)) * 3;
    int L, A, B;
    rgb2lab(T, q[0], q[1], q[2], L, A, B);
    atomicAdd(&hist[L], 1);
  }
  __syncthreads();
  const int area = ts * ts;
  int h = hist[t];                                 // (cl = max(int(clip * area / 256), 1), formed by the launcher in doubles)
  red[t] = h > cl ? h - cl : 0;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
  const int clipped = red[0];
  const int batch = clipped >> 8, residual = clipped & 255;
  h = (h < cl ? h : cl) + batch;
  if (residual) {
    const int step = 256 / residual > 1 ? 256 / residual : 1;
    if (t % step == 0 && t / step < residual) ++h;
  }
  __syncthreads();
  red[t] = h;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {              // inclusive scan (integers: order-free)
    const int v = t >= o ? red[t - o] : 0;
    __syncthreads();
    red[t] += v;
    __syncthreads();
  }
  const float scale = 255.0f / (float)area;
  luts[blockIdx.x * 256 + t] = (unsigned char)min(max((int)rintf((float)red[t] * scale), 0), 255);
}

__global__ __launch_bounds__(256) void clahe_apply_kernel(const unsigned char* src, unsigned char* dst, const unsigned char* luts,
                                                          const unsigned char* ctab, int S) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= S * S) return;
  const LabTab T = lab_tab(ctab);
  const int y = p / S, x = p % S;
  const unsigned char* q = src + (size_t)p * 3;
  int L, A, B;
  rgb2lab(T, q[0], q[1], q[2], L, A, B);
  const int ts = (S + 7) >> 3;
  const float inv = 1.0f / (float)ts;
  const float txf = (float)x * inv - 0.5f, tyf = (float)y * inv - 0.5f;
  const int tx1f = (int)floorf(txf), ty1f = (int)floorf(tyf);
  const float xa = txf - (float)tx1f, ya = tyf - (float)ty1f, xa1 = 1.0f - xa, ya1 = 1.0f - ya;
  const int tx1 = max(tx1f, 0), tx2 = min(tx1f + 1, 7), ty1 = max(ty1f, 0), ty2 = min(ty1f + 1, 7);
  const float l11 = luts[(ty1 * 8 + tx1) * 256 + L], l12 = luts[(ty1 * 8 + tx2) * 256 + L];
  const float l21 = luts[(ty2 * 8 + tx1) * 256 + L], l22 = luts[(ty2 * 8 + tx2) * 256 + L];
  const float res = (l11 * xa1 + l12 * xa) * ya1 + (l21 * xa1 + l22 * xa) * ya;
  int out[3];
  lab2rgb(T, min(max((int)rintf(res), 0), 255), A, B, out);
  dst[(size_t)p * 3 + 0] = (unsigned char)out[0]; dst[(size_t)p * 3 + 1] = (unsigned char)out[1]; dst[(size_t)p * 3 + 2] = (unsigned char)out[2];
}

}  // namespace

extern "C" {

int kodhip_compose_desc_bytes(void) { return (int)sizeof(SampleDesc); }

// The colour stage of ONE sample whose gate fired (image_color_transforms, kod/data/augmentations/default.py:420-432,
// 460-461): warps descriptor `entry` (= 2 * sample + slot of descs [B][2]) into out_u8 [S][S][3], then runs the fired
// transforms in the reference's Compose order - ops bit 0 Blur(blur_k), 1 MedianBlur(median_k), 2 ToGray, 3 CLAHE(clahe_clip,
// 8 x 8 tiles) - leaving the result in out_u8.  tmp_u8: a second S * S * 3 byte image; luts_u8: 64 * 256 bytes (CLAHE);
// color_tab: data/device_pipeline.color_table() on the device.  The caller sets that descriptor's `pre` to out_u8 BEFORE
// uploading descs, so that kodhip_compose_batch (launched after this on the same stream) reads the sample from there.
int kodhip_compose_color(const void* pool, const void* descs, const void* bilinear_tab, const void* color_tab, int entry,
                         int ops, int blur_k, int median_k, double clahe_clip, void* out_u8, void* tmp_u8, void* luts_u8,
                         int S, hipStream_t stream) {
  KOD_CHECK_ARG(pool && descs && bilinear_tab && color_tab && out_u8 && tmp_u8 && luts_u8 && entry >= 0 && S > 0, "compose_color: bad args");
  KOD_CHECK_ARG(ops > 0 && ops < 16, "compose_color: ops must name at least one of the four transforms");
  KOD_CHECK_ARG(!(ops & 1) || blur_k == 3 || blur_k == 5 || blur_k == 7, "compose_color: Blur kernel size %d", blur_k);
  KOD_CHECK_ARG(!(ops & 2) || median_k == 3 || median_k == 5 || median_k == 7, "compose_color: MedianBlur kernel size %d", median_k);
  KOD_CHECK_ARG(!(ops & 8) || (clahe_clip >= 1.0 && clahe_clip <= 4.0), "compose_color: CLAHE clip limit %g", clahe_clip);
  unsigned char* cur = (unsigned char*)out_u8;
  unsigned char* oth = (unsigned char*)tmp_u8;
  const dim3 g(cdiv((long)S * S, 256)), b(256);
  hipLaunchKernelGGL(warp_u8_kernel, g, b, 0, stream, (const unsigned char*)pool, (const SampleDesc*)descs, (const short*)bilinear_tab, entry, cur, S);
  KOD_LAUNCH_CHECK("compose_color warp");
  if (ops & 1) {
    hipLaunchKernelGGL(blur_u8_kernel, g, b, 0, stream, (const unsigned char*)cur, oth, S, blur_k);
    unsigned char* t = cur; cur = oth; oth = t;
  }
  if (ops & 2) {
    if (median_k == 3) hipLaunchKernelGGL(median_u8_kernel<3>, g, b, 0, stream, (const unsigned char*)cur, oth, S);
    else if (median_k == 5) hipLaunchKernelGGL(median_u8_kernel<5>, g, b, 0, stream, (const unsigned char*)cur, oth, S);
    else hipLaunchKernelGGL(median_u8_kernel<7>, g, b, 0, stream, (const unsigned char*)cur, oth, S);
    unsigned char* t = cur; cur = oth; oth = t;
  }
  if (ops & 4) hipLaunchKernelGGL(gray_u8_kernel, g, b, 0, stream, cur, S);
  if (ops & 8) {
    const int ts = (S + 7) / 8;
    int cl = (int)(clahe_clip * (double)(ts * ts) / 256.0);          // clahe.cpp: clipLimit * tileSizeTotal / histSize, at least 1
    if (cl < 1) cl = 1;
    hipLaunchKernelGGL(clahe_hist_kernel, dim3(64), b, 0, stream, (const unsigned char*)cur, (unsigned char*)luts_u8, (const unsigned char*)color_tab, S, cl);
    hipLaunchKernelGGL(clahe_apply_kernel, g, b, 0, stream, (const unsigned char*)cur, oth, (const unsigned char*)luts_u8, (const unsigned char*)color_tab, S);
    unsigned char* t = cur; cur = oth; oth = t;
  }
  KOD_LAUNCH_CHECK("compose_color ops");
  if (cur != (unsigned char*)out_u8) {
    hipError_t e = hipMemcpyAsync(out_u8, cur, (size_t)S * S * 3, hipMemcpyDeviceToDevice, stream);
    if (e != hipSuccess) { kodhip_set_error("compose_color: %s", hipGetErrorString(e)); return (int)e; }
  }
  return KOD_OK;
}

// pool: u8 source images (HWC) followed by at least 4 readable bytes (a pixel's three channels are fetched as one 4-byte load); descs: device [B][2] SampleDesc; mix: device [B][2] floats (r, 1-r) or (-1, 0);
// bilinear_tab: device 32*32*4 int16 (OpenCV fixed-point table) followed by OpenCV's sdiv_table[256] | hdiv_table180[256]
// (int32; data/device_pipeline.py bilinear_table()); out_f32 [B,3,S,S] and/or out_pairs (bf16).
int kodhip_compose_batch(const void* pool, const void* descs, const float* mix, const void* bilinear_tab,
                         float* out_f32, void* out_pairs, int B, int S, hipStream_t stream) {
  KOD_CHECK_ARG(pool && descs && mix && bilinear_tab && (out_f32 || out_pairs) && B > 0 && S > 0 && S % 2 == 0,
                "compose_batch: bad args");
  hipLaunchKernelGGL(compose_kernel, dim3(cdiv((long)S * S / 2, 256), B), dim3(256), 0, stream,
                     (const unsigned char*)pool, (const SampleDesc*)descs, mix, (const short*)bilinear_tab, out_f32,
                     (bf16_t*)out_pairs, S);
  KOD_LAUNCH_CHECK("compose_batch");
  return KOD_OK;
}

}  // extern "C"
