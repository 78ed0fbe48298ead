"""AugmentedSample (schema mirror of kod/data/types.py:8-11)."""
from __future__ import annotations

from typing import NamedTuple


class AugmentedSample(NamedTuple):
    image: object      # u8 HWC ndarray | DeviceCanvas (mosaic.py) | f32 CHW tensor after TrainSampleAugmentor
    bboxes: object     # [n, 4] xyxy pixels, numpy f64
    labels: object     # [n] numpy i64
