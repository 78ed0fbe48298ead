"""Prediction decode - Yolov5BoxPrediction / ObjectnessPrediction / ClassPrediction / Yolov5Prediction /
Yolov5PredictionAssembler (kod/lightning/experiments/yv5_baseline/layers.py:15-155) as classes of the same names over
the decode kernel, and DefaultYolov5Experiment.get_detections (exp.py:70-102) as ONE launch over the three head tensors
(what the validation loop uses)."""
from __future__ import annotations

from typing import Sequence

import torch
import torch.nn as nn

from .... import _lib
from ....core.types import FeatureShape
from .type_defs import LayerwiseAnchorInfo, PredictionResult


def _decode_level(raw: torch.Tensor, stride: int, anchors_wh: Sequence) -> torch.Tensor:
    """One pyramid level through the decode kernel: raw [B, A, h, w, 5+nc] fp32 -> [B, A*h*w, 5+nc]."""
    _lib.require_gpu()
    raw = raw.detach().contiguous().float()
    B, A, h, w, P = raw.shape
    levels = (_lib.KodDecodeLevel * 3)()
    for i in range(3):                      # levels 1, 2: empty (the kernel walks rows, not levels)
        lv = levels[i]
        lv.raw, lv.h, lv.w, lv.stride = raw.data_ptr(), (h if i == 0 else 0), (w if i == 0 else 0), stride
        for k, a in enumerate(anchors_wh):
            lv.anchor_w[k], lv.anchor_h[k] = float(a[0]), float(a[1])
    det = torch.empty((B, A * h * w, P), dtype=torch.float32, device=raw.device)
    _lib.check(_lib.lib().kodhip_decode(levels, det.data_ptr(), B, A, P - 5, torch.cuda.current_stream().cuda_stream), "decode")
    return det


class Yolov5BoxPrediction(nn.Module):
    """kod/lightning/experiments/yv5_baseline/layers.py:15-63: box logits [B, A, h, w, 4] -> xyxy pixels [B, A*h*w, 4]
    (xy = (sigmoid*2 + grid - .5) * stride, wh = (sigmoid*2)^2 * anchor; rows ordered anchor, y, x).  The input is left
    untouched (the reference overwrites it in place)."""

    def __init__(self, stride: int, image_feature_shape: FeatureShape, anchor_box_shapes: Sequence[FeatureShape]):
        super().__init__()
        self.stride = stride
        self.target_feature_shape = FeatureShape(width=image_feature_shape.width // stride,
                                                 height=image_feature_shape.height // stride)
        self.anchor_box_shapes = [(float(a[0]), float(a[1])) if not hasattr(a, "width") else (float(a.width), float(a.height))
                                  for a in anchor_box_shapes]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, A, h, w, _ = x.shape
        assert (h, w) == (self.target_feature_shape.height, self.target_feature_shape.width) and A == len(self.anchor_box_shapes)
        raw = torch.zeros((B, A, h, w, 6), dtype=torch.float32, device=x.device)       # one dummy class slot
        raw[..., :4] = x
        return _decode_level(raw, self.stride, self.anchor_box_shapes)[..., :4].contiguous()


class Yolov5ObjectnessPrediction(nn.Module):
    """layers.py:66-77: sigmoid, [B, A, h, w, 1] -> [B, A*h*w, 1]."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, A, h, w, _ = x.shape
        raw = torch.zeros((B, A, h, w, 6), dtype=torch.float32, device=x.device)
        raw[..., 4:5] = x
        return _decode_level(raw, 1, [(1.0, 1.0)] * A)[..., 4:5].contiguous()


class Yolov5ClassPrediction(nn.Module):
    """layers.py:80-91: sigmoid, [B, A, h, w, nc] -> [B, A*h*w, nc]."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, A, h, w, nc = x.shape
        raw = torch.zeros((B, A, h, w, 5 + nc), dtype=torch.float32, device=x.device)
        raw[..., 5:] = x
        return _decode_level(raw, 1, [(1.0, 1.0)] * A)[..., 5:].contiguous()


class Yolov5Prediction(nn.Module):
    """layers.py:94-128: one level's (box, objectness, class) head tensors -> PredictionResult, ONE decode launch."""

    def __init__(self, stride: int, image_feature_shape: FeatureShape, anchor_box_shapes: Sequence[FeatureShape]):
        super().__init__()
        self.box_prediction = Yolov5BoxPrediction(stride, image_feature_shape, anchor_box_shapes)
        self.obj_prediction = Yolov5ObjectnessPrediction()
        self.class_prediction = Yolov5ClassPrediction()

    def forward(self, box_features: torch.Tensor, obj_features: torch.Tensor, class_features: torch.Tensor) -> PredictionResult:
        raw = torch.cat((box_features, obj_features, class_features), -1)
        bp = self.box_prediction
        det = _decode_level(raw, bp.stride, bp.anchor_box_shapes)
        return PredictionResult(det[..., :4], det[..., 4:5], det[..., 5:])


class Yolov5PredictionAssembler(nn.Module):
    """layers.py:131-155: concatenates the levels along the row axis and (box, obj, cls) along the last: [B, rows, 5+nc]."""

    def forward(self, box_predictions: Sequence[torch.Tensor], obj_predictions: Sequence[torch.Tensor],
                class_predictions: Sequence[torch.Tensor]) -> torch.Tensor:
        return torch.cat((torch.cat(tuple(box_predictions), 1), torch.cat(tuple(obj_predictions), 1),
                          torch.cat(tuple(class_predictions), 1)), -1)


def get_detections(image_feature_shape: FeatureShape, net_result, anchor_info: LayerwiseAnchorInfo) -> torch.Tensor:
    """[B, sum(A*h*w), 5+nc]: xyxy pixels, sigmoid(obj), sigmoid(cls); rows ordered (level, anchor, y, x)."""
    from .loss import Yolov5Loss
    _lib.require_gpu()
    raws = [Yolov5Loss._raw(h).detach().contiguous() for h in net_result]
    B, A, _, _, P = raws[0].shape
    levels = (_lib.KodDecodeLevel * 3)()
    rows = 0
    for i, (t, info) in enumerate(zip(raws, anchor_info)):
        lv = levels[i]
        lv.raw, lv.h, lv.w, lv.stride = t.data_ptr(), t.shape[2], t.shape[3], info.stride
        assert t.shape[2] == image_feature_shape.height // info.stride and t.shape[3] == image_feature_shape.width // info.stride
        for k, a in enumerate(info.boxes_wh):
            lv.anchor_w[k], lv.anchor_h[k] = float(a.width), float(a.height)
        rows += A * t.shape[2] * t.shape[3]
    det = torch.empty((B, rows, P), dtype=torch.float32, device=raws[0].device)
    _lib.check(_lib.lib().kodhip_decode(levels, det.data_ptr(), B, A, P - 5,
                                        torch.cuda.current_stream().cuda_stream), "decode")
    return det
