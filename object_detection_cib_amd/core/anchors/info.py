"""Mirrors kod/core/anchors/info.py:9-11; default anchors = kod/configs/anchor_boxes/voc_s{8,16,32}.yaml."""
from __future__ import annotations

from typing import NamedTuple, Sequence

from ..types import FeatureShape


class AnchorBoxInfo(NamedTuple):
    stride: int
    boxes_wh: Sequence[FeatureShape]


def voc_anchor_info(stride: int) -> AnchorBoxInfo:
    table = {8: ((10, 13), (16, 30), (33, 23)), 16: ((30, 61), (62, 45), (59, 119)),
             32: ((116, 90), (156, 198), (373, 326))}
    return AnchorBoxInfo(stride=stride, boxes_wh=[FeatureShape(width=w, height=h) for w, h in table[stride]])
