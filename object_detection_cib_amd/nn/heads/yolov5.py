"""Yolov5Head - drop-in for kod.nn.heads.yolov5.Yolov5Head (kod/nn/heads/yolov5.py:139-178): the box / objectness / class
1x1 convs of one pyramid level as ONE GEMM (N = A * (5 + nc)), parameters box_head.conv.*, obj_head.conv.*,
cls_head.conv.* with the reference's bias initialisation."""
from __future__ import annotations

import torch

from ...engine.graph import build_head_graph
from ..graph_module import GraphModule
from .types import DetectionHeadResult


class Yolov5Head(GraphModule):
    def __init__(self, in_channels: int, num_anchors_per_cell: int, num_classes: int, stride: int,
                 prior_probability: float = 0.01, use_yv5_init: bool = True):
        super().__init__()
        # (heads/yolov5.py:65-73,113-121: the YOLOv5 bias shifts by default, the focal-loss prior of `prior_probability` otherwise)
        self._init_graph(build_head_graph(in_channels, num_anchors_per_cell, num_classes, stride), None, use_yv5_init, prior_probability)

    def forward(self, x: torch.Tensor) -> DetectionHeadResult:
        t = self._run([x])[0][0]                  # [B, A, h, w, 5 + nc]: the reference's 'b (a p) h w -> b a h w p' views
        return DetectionHeadResult(t[..., 0:4], t[..., 4:5], t[..., 5:])
