// Device data path: mosaic paste + affine warp + HSV jitter + flip + /255 (+ mixup) in ONE gather kernel.
//
// Reference (per sample, in DataLoader workers, numpy / OpenCV):
//   mosaic   kod/data/mosaic.py:58-132          4 u8 HWC images pasted on a 2S x 2S canvas filled with 114
//   affine   kod/data/augmentations/default.py:279-320   cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT 114) -> S x S
//            (or cv2.warpPerspective when a perspective draw is non-zero, default.py:306-313)
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

// one composite pixel (3 channels, u8 domain) of sample d at output (y, x)
__device__ __forceinline__ void composite(const unsigned char* pool, const SampleDesc& d, const short* tab,
                                          const int* sdiv_tab, const int* hdiv_tab, int y, int xo, int S, int out[3]) {
  const int x = d.flip ? S - 1 - xo : xo;
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

}  // namespace

extern "C" {

int kodhip_compose_desc_bytes(void) { return (int)sizeof(SampleDesc); }

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
