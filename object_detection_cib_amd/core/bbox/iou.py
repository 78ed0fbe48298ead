"""IoUCalculator / compute_* - drop-ins for kod.core.bbox.iou (kod/core/bbox/iou.py:9-14,77-95,142-268).

Same enum, constructor and call: ``IoUCalculator(IoUType.ciou, eps)(boxes1, boxes2) -> [m]`` on aligned xyxy rows,
differentiable w.r.t. both box tensors.  The arithmetic is csrc/iou.hip (forward + a forward-mode backward that
follows autograd's tie / clamp conventions); there is no CPU path.  The training loss keeps its own fused CIoU
(csrc/loss.hip) - Yolov5Loss accepts the calculator for signature parity and checks it asks for (ciou, 1e-7).
"""
from __future__ import annotations

import enum

import torch

from ... import _lib


@enum.unique
class IoUType(str, enum.Enum):
    ioU = "iou"
    giou = "giou"
    diou = "diou"
    ciou = "ciou"


_KIND = {IoUType.ioU: 0, IoUType.giou: 1, IoUType.diou: 2, IoUType.ciou: 3}


class _IoUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, b1, b2, kind, eps):
        _lib.require_gpu()
        if not (b1.is_cuda and b2.is_cuda):
            raise RuntimeError("IoUCalculator (HIP) needs CUDA tensors; there is no CPU fallback")
        shape = b1.shape[:-1]
        a = b1.detach().reshape(-1, 4).contiguous().float()
        b = b2.detach().reshape(-1, 4).contiguous().float()
        out = torch.empty(a.shape[0], dtype=torch.float32, device=a.device)
        _lib.check(_lib.lib().kodhip_iou_fwd(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.shape[0], kind, eps,
                                             torch.cuda.current_stream().cuda_stream), "iou_fwd")
        ctx.save_for_backward(a, b)
        ctx.kind, ctx.eps, ctx.in_shape = kind, eps, (b1.shape, b2.shape)
        return out.reshape(shape)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        need1, need2 = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        g = g.reshape(-1).contiguous().float()
        g1 = torch.empty_like(a) if need1 else None
        g2 = torch.empty_like(b) if need2 else None
        _lib.check(_lib.lib().kodhip_iou_bwd(a.data_ptr(), b.data_ptr(), g.data_ptr(),
                                             g1.data_ptr() if need1 else None, g2.data_ptr() if need2 else None,
                                             a.shape[0], ctx.kind, ctx.eps,
                                             torch.cuda.current_stream().cuda_stream), "iou_bwd")
        return (g1.reshape(ctx.in_shape[0]) if need1 else None, g2.reshape(ctx.in_shape[1]) if need2 else None,
                None, None)


def _call(kind: IoUType, boxes1: torch.Tensor, boxes2: torch.Tensor, eps: float) -> torch.Tensor:
    assert boxes1.shape == boxes2.shape and boxes1.shape[-1] == 4
    return _IoUFn.apply(boxes1, boxes2, _KIND[kind], float(eps))


def compute_iou(boxes1, boxes2, eps: float = 1e-7):
    return _call(IoUType.ioU, boxes1, boxes2, eps)


def compute_giou(boxes1, boxes2, eps: float = 1e-7):
    return _call(IoUType.giou, boxes1, boxes2, eps)


def compute_diou(boxes1, boxes2, eps: float = 1e-7):
    return _call(IoUType.diou, boxes1, boxes2, eps)


def compute_ciou(boxes1, boxes2, eps: float = 1e-7):
    return _call(IoUType.ciou, boxes1, boxes2, eps)


class IoUCalculator(object):
    def __init__(self, iou_type: IoUType = IoUType.ciou, eps: float = 1e-7):
        self.iou_type = IoUType(iou_type)
        self.eps = eps
        self.fn = {IoUType.ioU: compute_iou, IoUType.giou: compute_giou, IoUType.diou: compute_diou,
                   IoUType.ciou: compute_ciou}[self.iou_type]

    def __call__(self, boxes1: torch.Tensor, boxes2: torch.Tensor):
        return self.fn(boxes1, boxes2, self.eps)
