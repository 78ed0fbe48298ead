"""IoU selection object (mirrors kod/core/bbox/iou.py:9-14,249-268).  On the HIP path the IoU lives inside
the fused loss kernel (csrc/loss.hip); only the reference's configured type, CIoU, is implemented there."""
from __future__ import annotations

import enum


@enum.unique
class IoUType(str, enum.Enum):
    ioU = "iou"
    giou = "giou"
    diou = "diou"
    ciou = "ciou"


class IoUCalculator(object):
    def __init__(self, iou_type: IoUType = IoUType.ciou, eps: float = 1e-7):
        self.iou_type = IoUType(iou_type)
        self.eps = eps
        if self.iou_type is not IoUType.ciou or abs(eps - 1e-7) > 1e-12:
            raise NotImplementedError("HIP loss kernel implements iou_type=ciou, eps=1e-7 "
                                      "(kod/configs/nn/losses/yv5.yaml:13-16)")
