"""Prediction decode - replaces Yolov5BoxPrediction / ObjectnessPrediction / ClassPrediction /
Yolov5PredictionAssembler (kod/lightning/experiments/yv5_baseline/layers.py:15-155) and
DefaultYolov5Experiment.get_detections (exp.py:70-102) with one HIP kernel over the three head tensors."""
from __future__ import annotations

import torch

from .... import _lib
from ....core.types import FeatureShape
from .type_defs import LayerwiseAnchorInfo


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
