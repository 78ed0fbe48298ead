"""Batch element types (mirror kod/data/detection.py:24-37)."""
from __future__ import annotations

from typing import NamedTuple, Optional

import torch


class DetectionTarget(NamedTuple):
    boxes: torch.Tensor     # [n,4] xyxy pixels, float64 in the reference
    labels: torch.Tensor    # [n] int64


class DetectionSample(NamedTuple):
    img: torch.Tensor
    target: DetectionTarget
    image_info: Optional[object] = None
