"""Yolov5LabelAssigner - drop-in for kod.core.label_assignment.yv5 (kod/core/label_assignment/yv5.py:18-319).

Same constructor / call signature and result tuples; the ~40 small aten ops per pyramid level of the
reference are one HIP kernel (csrc/loss.hip::assign_kernel) that reproduces the reference's row order
bit for bit.  ``assign_device`` is the sync-free form used by the fused loss (row counts stay on the GPU).
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple, Sequence

import torch

from ... import _lib
from ..types import FeatureShape
from ..anchors.info import AnchorBoxInfo


class AssignmentAnchorInfo(NamedTuple):
    ll: AnchorBoxInfo
    ml: AnchorBoxInfo
    hl: AnchorBoxInfo


class AssignmentTargetIndices(NamedTuple):
    samples: torch.Tensor
    anchors: torch.Tensor
    grid_y: torch.Tensor
    grid_x: torch.Tensor


class AssignmentTargetInfo(NamedTuple):
    indices: AssignmentTargetIndices
    labels: torch.Tensor
    gt_boxes: torch.Tensor
    anchors: torch.Tensor
    target_feature_shape: FeatureShape


class AssignmentResult(NamedTuple):
    ll: AssignmentTargetInfo
    ml: AssignmentTargetInfo
    hl: AssignmentTargetInfo


class DeviceAssignment(NamedTuple):
    """Capacity-sized device buffers of one level (first `count` rows valid)."""
    idx: torch.Tensor       # int32 [4, cap]  sample, anchor, grid_y, grid_x
    label: torch.Tensor     # int32 [cap]
    gt: torch.Tensor        # f32 [cap, 4]
    anc: torch.Tensor       # f32 [cap, 2]
    count: torch.Tensor     # int32 [1]
    stride: int


class BatchedTargets(NamedTuple):
    """All boxes of a batch, already concatenated on the device (what the collate function of a device data
    pipeline hands over; also makes the train step capturable in a hipGraph: no host tensors involved)."""
    boxes: torch.Tensor      # f64 [n, 4] xyxy pixels
    labels: torch.Tensor     # i64 [n]
    samples: torch.Tensor    # i32 [n] image index of every box
    n: int

    @staticmethod
    def from_targets(targets, device) -> "BatchedTargets":
        lens = [int(t.boxes.shape[0]) for t in targets]
        n = sum(lens)
        if n == 0:
            z = torch.zeros((0, 4), dtype=torch.float64, device=device)
            return BatchedTargets(z, torch.zeros(0, dtype=torch.int64, device=device),
                                  torch.zeros(0, dtype=torch.int32, device=device), 0)
        boxes = torch.cat([t.boxes.reshape(-1, 4).to(torch.float64) for t in targets], 0)
        labels = torch.cat([t.labels.reshape(-1).to(torch.int64) for t in targets], 0)
        samples = torch.repeat_interleave(torch.arange(len(lens), dtype=torch.int32),
                                          torch.tensor(lens, dtype=torch.int64))
        return BatchedTargets(boxes.to(device).contiguous(), labels.to(device).contiguous(), samples.to(device), n)


class Yolov5LabelAssigner(object):
    def __init__(self, anchor_info: AssignmentAnchorInfo, threshold: float = 4.0):
        self.anchor_info = anchor_info
        self.threshold = threshold
        self.off_bias = 0.5
        for a in anchor_info:
            if len(a.boxes_wh) != 3:
                raise ValueError("HIP assigner kernel is built for 3 anchors per cell")

    # -- sync-free device path ------------------------------------------------------------------
    def assign_device(self, input_image_shape: FeatureShape, targets, device, stream=None) -> tuple:
        """stream: torch stream the assignment kernel is launched on (default: the current one).  The outputs are
        allocated in the current stream's context either way; a caller that passes a side stream joins it before use
        (the assignment depends only on the targets, so a training step runs it beside the network's forward pass)."""
        _lib.require_gpu()
        lib = _lib.lib()
        bt = targets if isinstance(targets, BatchedTargets) else BatchedTargets.from_targets(targets, device)
        n = bt.n
        cap = max(15 * n, 16)
        boxes, labels, samples = bt.boxes, bt.labels, bt.samples
        levels = (_lib.KodAssignLevel * 3)()
        outs = []
        for i, info in enumerate(self.anchor_info):
            d = DeviceAssignment(idx=torch.empty((4, cap), dtype=torch.int32, device=device),
                                 label=torch.empty(cap, dtype=torch.int32, device=device),
                                 gt=torch.empty((cap, 4), dtype=torch.float32, device=device),
                                 anc=torch.empty((cap, 2), dtype=torch.float32, device=device),
                                 count=torch.empty(1, dtype=torch.int32, device=device), stride=info.stride)
            lv = levels[i]
            lv.idx, lv.label, lv.gt, lv.anc, lv.count = (d.idx.data_ptr(), d.label.data_ptr(), d.gt.data_ptr(),
                                                         d.anc.data_ptr(), d.count.data_ptr())
            for k, a in enumerate(info.boxes_wh):
                # python-float arithmetic then fp32, as torch.tensor(scaled_anchor_boxes) does (yv5.py:226-238)
                lv.anchor_w[k] = a.width * 1 / info.stride
                lv.anchor_h[k] = a.height * 1 / info.stride
            lv.stride = info.stride
            outs.append(d)
        _lib.check(lib.kodhip_assign_targets(boxes.data_ptr() if n else None, labels.data_ptr() if n else None,
                                             samples.data_ptr() if n else None, n, cap,
                                             int(input_image_shape.width), int(input_image_shape.height),
                                             float(self.threshold), levels,
                                             (stream or torch.cuda.current_stream()).cuda_stream), "assign_targets")
        self._keepalive = (boxes, labels, samples)
        return tuple(outs), cap

    # -- reference-shaped API (synchronises to size the outputs) ----------------------------------
    def __call__(self, input_image_shape: FeatureShape, targets: Sequence) -> AssignmentResult:
        device = torch.device("cuda", torch.cuda.current_device())
        for t in targets:
            if t.boxes.is_cuda:
                device = t.boxes.device
                break
        levels, _ = self.assign_device(input_image_shape, targets, device)
        res = []
        for d in levels:
            m = int(d.count.item())
            idx = d.idx[:, :m].long()
            res.append(AssignmentTargetInfo(
                indices=AssignmentTargetIndices(samples=idx[0], anchors=idx[1], grid_y=idx[2], grid_x=idx[3]),
                labels=d.label[:m].long(), gt_boxes=d.gt[:m].clone(), anchors=d.anc[:m].clone(),
                target_feature_shape=FeatureShape(width=input_image_shape.width // d.stride,
                                                  height=input_image_shape.height // d.stride)))
        return AssignmentResult(*res)
