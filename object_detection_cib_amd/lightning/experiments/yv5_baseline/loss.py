"""Yolov5Loss - drop-in for kod.lightning.experiments.yv5_baseline.loss (loss.py:25-248).

Same constructor / call signature / LossResult; assignment + gather + decode + CIoU + the three BCE terms
and their analytic backward run as a handful of HIP kernels (csrc/loss.hip).  Reference quirks kept:
non-detached objectness target, last-writer-wins on duplicate cells, NaN on an empty level, per-level
means, lambda_obj*(W/640)^2, lambda_cls*nc/80.
"""
from __future__ import annotations

from typing import NamedTuple, Sequence

import torch
import torch.nn as nn

from .... import _lib
from ....core.types import FeatureShape
from ....core.bbox.iou import IoUCalculator
from ....core.label_assignment.yv5 import Yolov5LabelAssigner
from .type_defs import LossResult


class Yolov5LossParams(NamedTuple):
    lambda_classification: float
    lambda_localization: float
    lambda_objectness: float
    lambda_ll_objectness: float
    lambda_ml_objectness: float
    lambda_hl_objectness: float

    @staticmethod
    def get_default() -> "Yolov5LossParams":
        return Yolov5LossParams(0.5, 0.05, 1.0, 4.0, 1.0, 0.4)


def _run(mod, shape, raws, asg, cap, grads, upstream, work):
    lib = _lib.lib()
    B, A, _, _, P = raws[0].shape
    nc = P - 5
    hp = mod.hparams
    levels = (_lib.KodLossLevel * 3)()
    for i, (t, d, bal) in enumerate(zip(raws, asg, (hp.lambda_ll_objectness, hp.lambda_ml_objectness,
                                                    hp.lambda_hl_objectness))):
        lv = levels[i]
        lv.logits = t.data_ptr()
        lv.grad = grads[i].data_ptr() if grads is not None else None
        lv.idx, lv.label, lv.gt, lv.anc, lv.count = (d.idx.data_ptr(), d.label.data_ptr(), d.gt.data_ptr(),
                                                     d.anc.data_ptr(), d.count.data_ptr())
        lv.cellmaps, lv.rowprev, lv.rowgrad, lv.tobj = (work["maps"][i].data_ptr(), work["prev"][i].data_ptr(),
                                                        work["rowgrad"][i].data_ptr(), work["tobj"][i].data_ptr())
        lv.fh, lv.fw, lv.balance = t.shape[2], t.shape[3], bal
    lam_obj = hp.lambda_objectness * ((shape.width / 640) ** 2)          # loss.py:231-233
    lam_cls = hp.lambda_classification * (nc / 80)                        # loss.py:235-237
    pw = mod.weights
    _lib.check(lib.kodhip_yolo_loss_iou(levels, B, A, nc, cap, hp.lambda_localization, lam_obj, lam_cls,
                                        pw.data_ptr() if pw is not None else None,
                                        upstream.data_ptr() if upstream is not None else None,
                                        work["partials"].data_ptr(), work["nslots"], work["out"].data_ptr(),
                                        1 if grads is not None else 0, mod.iou_kind, mod.iou_eps,
                                        torch.cuda.current_stream().cuda_stream), "yolo_loss")


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, shape, targets, t_ll, t_ml, t_hl):
        raws = [t.contiguous() for t in (t_ll, t_ml, t_hl)]
        dev = raws[0].device
        asg, cap = mod.assigner.assign_device(shape, targets, dev)
        B, A, _, _, P = raws[0].shape
        nslots = max((cap + 255) // 256, 1024)
        work = dict(
            maps=[torch.empty(3 * B * A * t.shape[2] * t.shape[3], dtype=torch.int32, device=dev) for t in raws],
            prev=[torch.empty(cap, dtype=torch.int32, device=dev) for _ in raws],
            rowgrad=[torch.empty(cap * (P - 1), dtype=torch.float32, device=dev) for _ in raws],
            tobj=[torch.empty(cap, dtype=torch.float32, device=dev) for _ in raws],
            partials=torch.empty(9 * nslots, dtype=torch.float32, device=dev), nslots=nslots,
            out=torch.empty(16, dtype=torch.float32, device=dev))
        _run(mod, shape, raws, asg, cap, None, None, work)
        ctx.state = (mod, shape, raws, asg, cap, work)
        out = work["out"]
        return out[0].clone(), out[1].clone(), out[2].clone()

    @staticmethod
    def backward(ctx, g_box, g_obj, g_cls):
        mod, shape, raws, asg, cap, work = ctx.state
        upstream = torch.stack([g_box, g_obj, g_cls]).to(torch.float32).contiguous()
        grads = [torch.empty_like(t) for t in raws]
        work = dict(work)
        work["out"] = torch.empty(16, dtype=torch.float32, device=raws[0].device)
        _run(mod, shape, raws, asg, cap, grads, upstream, work)
        return (None, None, None, *grads)


class Yolov5Loss(nn.Module):
    def __init__(self, assigner: Yolov5LabelAssigner, hparams: Yolov5LossParams,
                 iou_calculator: IoUCalculator, weights: list[float] = None):
        super().__init__()
        self.assigner = assigner
        self.hparams = hparams
        self.iou_calculator = iou_calculator
        # loss.py:46-63 takes any IoUCalculator; the kernel evaluates its kind / eps: ciou with eps 1e-7
        # (kod/configs/nn/losses/yv5.yaml:13-16) in closed form, the other members of the family on dual numbers (csrc/kodhip_iou.h)
        kind = getattr(getattr(iou_calculator, "iou_type", None), "value", getattr(iou_calculator, "iou_type", None))
        kinds = {"iou": 0, "giou": 1, "diou": 2, "ciou": 3}
        if kind not in kinds:
            raise ValueError(f"unknown IoU type {kind!r}: expected one of {sorted(kinds)} (kod/core/bbox/iou.py:9-14)")
        self.iou_kind = kinds[kind]
        self.iou_eps = float(getattr(iou_calculator, "eps", 1e-7))
        # reference: plain attribute moved to CUDA when available (loss.py:58-61)
        self.weights = torch.tensor(weights, dtype=torch.float32) if weights is not None else None

    @staticmethod
    def _raw(head) -> torch.Tensor:
        """[B,A,h,w,5+nc] tensor behind a DetectionHeadResult (zero-copy when it came from Yolov5Network)."""
        box, obj, cls = head
        base = box._base
        if (base is not None and obj._base is base and cls._base is base and base.dim() == 5
                and base.shape[-1] == 5 + cls.shape[-1] and base.is_contiguous()
                and box.storage_offset() == base.storage_offset()):
            return base
        return torch.cat((box, obj, cls), -1)

    def forward(self, image_feature_shape: FeatureShape, net_result, targets: Sequence) -> LossResult:
        raws = [self._raw(h) for h in net_result]
        if self.weights is not None and self.weights.device != raws[0].device:
            self.weights = self.weights.to(raws[0].device)
        loc, obj, cls = _LossFn.apply(self, image_feature_shape, targets, *raws)
        return LossResult(localization=loc, objectness=obj, classification=cls)

    def value_and_grad(self, image_feature_shape: FeatureShape, raws, targets: Sequence,
                       upstream: Sequence[float] = (1.0, 1.0, 1.0), assignment=None):
        """Loss values AND d(sum_k upstream_k * loss_k) / d(head tensors) in ONE pass of the loss kernels - what
        `forward()` followed by autograd's backward computes in two passes (the second pass recomputes every row and
        cell).  raws: the three contiguous [B, A, h, w, 5+nc] fp32 head tensors; upstream = (d total / d localization,
        d total / d objectness, d total / d classification), e.g. (B, B, B) for the reference's `B * (loc + cls + obj)`
        (exp.py:104-121).  Same kernels, same arithmetic, same results bit for bit (tests/test_hip_training.py).
        assignment: a precomputed `assigner.assign_device(...)` result (a training step computes it on a side stream
        while the network's forward pass runs).  Returns (LossResult, [grad_ll, grad_ml, grad_hl])."""
        raws = [t.contiguous() for t in raws]
        dev = raws[0].device
        if self.weights is not None and self.weights.device != dev:
            self.weights = self.weights.to(dev)
        key = (tuple(float(u) for u in upstream), dev)
        cache = self.__dict__.setdefault("_upstream_cache", {})
        if key not in cache:          # (first call = warm-up, outside any graph capture)
            cache[key] = torch.tensor(key[0], dtype=torch.float32, device=dev)
        asg, cap = assignment if assignment is not None else self.assigner.assign_device(image_feature_shape, targets, dev)
        B, A, _, _, P = raws[0].shape
        nslots = max((cap + 255) // 256, 1024)
        work = dict(
            maps=[torch.empty(3 * B * A * t.shape[2] * t.shape[3], dtype=torch.int32, device=dev) for t in raws],
            prev=[torch.empty(cap, dtype=torch.int32, device=dev) for _ in raws],
            rowgrad=[torch.empty(cap * (P - 1), dtype=torch.float32, device=dev) for _ in raws],
            tobj=[torch.empty(cap, dtype=torch.float32, device=dev) for _ in raws],
            partials=torch.empty(9 * nslots, dtype=torch.float32, device=dev), nslots=nslots,
            out=torch.empty(16, dtype=torch.float32, device=dev))
        grads = [torch.empty_like(t) for t in raws]
        _run(self, image_feature_shape, raws, asg, cap, grads, cache[key], work)
        out = work["out"]
        # out[12] = upstream[0] * ((localization + classification) + objectness), written by the loss kernel: the step's
        # scalar when the three upstreams are one scale (Yolov5Network.train_step)
        self._scaled_total = out[12] if key[0][0] == key[0][1] == key[0][2] else None
        return LossResult(localization=out[0], objectness=out[1], classification=out[2]), grads
