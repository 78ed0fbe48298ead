// Dual-number evaluation of one aligned IoU / GIoU / DIoU / CIoU row (kod/core/bbox/iou.py:77-95,142-246): value and the 8
// partial derivatives (d/d boxes1[0..3], d/d boxes2[0..3]) in one pass, following autograd's conventions where the
// function is not smooth (ties of max / min split evenly, clamp(0) passes at 0, abs -> sign, CIoU's alpha a constant).
// Shared by csrc/iou.hip (IoUCalculator as an op) and csrc/loss.hip (Yolov5Loss with an iou_type other than its fused CIoU).
#pragma once
#include "kodhip_common.h"

namespace {

enum { KIND_IOU = 0, KIND_GIOU = 1, KIND_DIOU = 2, KIND_CIOU = 3 };

template <int N>
struct Dual {
  float v;
  float d[N > 0 ? N : 1];
};

template <int N> __device__ __forceinline__ Dual<N> cst(float v) {
  Dual<N> r; r.v = v;
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = 0.f;
  return r;
}
template <int N> __device__ __forceinline__ Dual<N> var(float v, int idx) {
  Dual<N> r = cst<N>(v);
  if (idx < N) r.d[idx] = 1.f;
  return r;
}
template <int N> __device__ __forceinline__ Dual<N> operator+(const Dual<N>& a, const Dual<N>& b) {
  Dual<N> r; r.v = a.v + b.v;
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i];
  return r;
}
template <int N> __device__ __forceinline__ Dual<N> operator-(const Dual<N>& a, const Dual<N>& b) {
  Dual<N> r; r.v = a.v - b.v;
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i];
  return r;
}
template <int N> __device__ __forceinline__ Dual<N> operator*(const Dual<N>& a, const Dual<N>& b) {
  Dual<N> r; r.v = a.v * b.v;
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
  return r;
}
template <int N> __device__ __forceinline__ Dual<N> operator/(const Dual<N>& a, const Dual<N>& b) {
  Dual<N> r; r.v = a.v / b.v;
  const float inv = 1.0f / b.v;
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
  return r;
}
template <int N> __device__ __forceinline__ Dual<N> addc(const Dual<N>& a, float c) { Dual<N> r = a; r.v += c; return r; }
template <int N> __device__ __forceinline__ Dual<N> mulc(const Dual<N>& a, float c) {
  Dual<N> r; r.v = a.v * c;
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * c;
  return r;
}
// torch.max(a, b) / torch.min(a, b) (aten maximum / minimum): the larger (smaller) operand gets the gradient, ties 1/2 each
template <int N> __device__ __forceinline__ Dual<N> dmax(const Dual<N>& a, const Dual<N>& b) {
  if (a.v > b.v) return a;
  if (b.v > a.v) return b;
  return mulc(a + b, 0.5f);
}
template <int N> __device__ __forceinline__ Dual<N> dmin(const Dual<N>& a, const Dual<N>& b) {
  if (a.v < b.v) return a;
  if (b.v < a.v) return b;
  return mulc(a + b, 0.5f);
}
template <int N> __device__ __forceinline__ Dual<N> clamp0(const Dual<N>& a) {     // x.clamp(0): grad where x >= 0
  if (a.v >= 0.f) return a;
  return cst<N>(0.f);
}
template <int N> __device__ __forceinline__ Dual<N> dabs(const Dual<N>& a) {
  const float s = a.v > 0.f ? 1.f : (a.v < 0.f ? -1.f : 0.f);
  Dual<N> r = mulc(a, s);
  r.v = fabsf(a.v);
  return r;
}
template <int N> __device__ __forceinline__ Dual<N> datan(const Dual<N>& a) {
  Dual<N> r; r.v = atanf(a.v);
  const float g = 1.0f / (1.0f + a.v * a.v);
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * g;
  return r;
}

// one row; N = 0 (value only: every derivative loop vanishes) or 8
template <int N>
__device__ __forceinline__ Dual<N> iou_row(const float* p1, const float* p2, int kind, float eps) {
  typedef Dual<N> T;
  const T x1 = var<N>(p1[0], 0), y1 = var<N>(p1[1], 1), x2 = var<N>(p1[2], 2), y2 = var<N>(p1[3], 3);
  const T x1g = var<N>(p2[0], 4), y1g = var<N>(p2[1], 5), x2g = var<N>(p2[2], 6), y2g = var<N>(p2[3], 7);
  // _intersection_area / _union_area (iou.py:39-64)
  const T inter = clamp0(dmin(x2, x2g) - dmax(x1, x1g)) * clamp0(dmin(y2, y2g) - dmax(y1, y1g));
  const T uni = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - inter;
  const T iou = inter / addc(uni, eps);
  if (kind == KIND_IOU) return iou;
  // _convex_width_height (iou.py:67-74)
  const T cw = dmax(x2, x2g) - dmin(x1, x1g);
  const T ch = dmax(y2, y2g) - dmin(y1, y1g);
  if (kind == KIND_GIOU) {
    const T ca = cw * ch;
    return iou - dabs(ca - uni) / dabs(addc(ca, eps));
  }
  const T diag = cw * cw + ch * ch;
  const T dx = mulc(x1 + x2, 0.5f) - mulc(x1g + x2g, 0.5f);
  const T dy = mulc(y1 + y2, 0.5f) - mulc(y1g + y2g, 0.5f);
  const T D = (dx * dx + dy * dy) / addc(diag, eps);
  if (kind == KIND_DIOU) return iou - D;
  const T w1 = x2 - x1, h1 = y2 - y1, w2 = x2g - x1g, h2 = y2g - y1g;
  const T da = datan(w2 / addc(h2, eps)) - datan(w1 / addc(h1, eps));
  const T v = mulc(da * da, 4.0f / (3.14159265358979323846f * 3.14159265358979323846f));
  const float alpha = v.v / ((1.0f - iou.v) + v.v + eps);           // no_grad (iou.py:238-239)
  return iou - D - mulc(v, alpha);
}

}  // namespace
