// Aligned IoU family (IoU / GIoU / DIoU / CIoU), forward and backward - the arithmetic behind
// kod.core.bbox.iou.IoUCalculator.__call__ (kod/core/bbox/iou.py:77-95 compute_iou, :142-168 compute_giou,
// :171-197 compute_diou, :200-246 compute_ciou, :249-268 IoUCalculator).  boxes1 / boxes2 are [m][4] xyxy fp32, the
// result is [m].  The training loss has its own fused CIoU (csrc/loss.hip); this op is the stand-alone form of
// the reference's public IoU API.
//
// Backward is forward-mode: each thread re-evaluates its row on dual numbers carrying the 8 partial derivatives
// (d/d boxes1[0..3], d/d boxes2[0..3]), following autograd's conventions where the function is not smooth:
// torch.max / torch.min of two tensors split the gradient evenly on ties, clamp(0) passes the gradient at exactly
// 0, abs has derivative sign(x) (0 at 0), and CIoU's alpha is a constant (computed under no_grad, iou.py:238-239).
#include "kodhip_common.h"
#include "kodhip_iou.h"

namespace {

__global__ void iou_fwd_kernel(const float* b1, const float* b2, float* out, long m, int kind, float eps) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  out[i] = iou_row<0>(b1 + 4 * i, b2 + 4 * i, kind, eps).v;
}

__global__ void iou_bwd_kernel(const float* b1, const float* b2, const float* gout, float* g1, float* g2, long m,
                               int kind, float eps) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const Dual<8> r = iou_row<8>(b1 + 4 * i, b2 + 4 * i, kind, eps);
  const float g = gout[i];
  if (g1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) g1[4 * i + k] = g * r.d[k];
  }
  if (g2) {
#pragma unroll
    for (int k = 0; k < 4; ++k) g2[4 * i + k] = g * r.d[4 + k];
  }
}

}  // namespace

extern "C" {

// kind: 0 iou, 1 giou, 2 diou, 3 ciou (IoUType order, kod/core/bbox/iou.py:9-14)
int kodhip_iou_fwd(const float* boxes1, const float* boxes2, float* out, long m, int kind, float eps,
                   hipStream_t stream) {
  KOD_CHECK_ARG(m >= 0 && kind >= 0 && kind <= 3, "iou_fwd: bad args (m=%ld kind=%d)", m, kind);
  if (m == 0) return KOD_OK;
  KOD_CHECK_ARG(boxes1 && boxes2 && out, "iou_fwd: null pointer");
  hipLaunchKernelGGL(iou_fwd_kernel, dim3(cdiv(m, 256)), dim3(256), 0, stream, boxes1, boxes2, out, m, kind, eps);
  KOD_LAUNCH_CHECK("iou_fwd");
  return KOD_OK;
}

// grad_boxes1 / grad_boxes2: [m][4] out, either may be NULL; grad_out [m]
int kodhip_iou_bwd(const float* boxes1, const float* boxes2, const float* grad_out, float* grad_boxes1,
                   float* grad_boxes2, long m, int kind, float eps, hipStream_t stream) {
  KOD_CHECK_ARG(m >= 0 && kind >= 0 && kind <= 3, "iou_bwd: bad args (m=%ld kind=%d)", m, kind);
  if (m == 0) return KOD_OK;
  KOD_CHECK_ARG(boxes1 && boxes2 && grad_out && (grad_boxes1 || grad_boxes2), "iou_bwd: null pointer");
  hipLaunchKernelGGL(iou_bwd_kernel, dim3(cdiv(m, 256)), dim3(256), 0, stream, boxes1, boxes2, grad_out,
                     grad_boxes1, grad_boxes2, m, kind, eps);
  KOD_LAUNCH_CHECK("iou_bwd");
  return KOD_OK;
}

}  // extern "C"
