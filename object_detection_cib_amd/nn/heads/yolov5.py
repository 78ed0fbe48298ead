"""Yolov5Head - drop-in for kod.nn.heads.yolov5.Yolov5Head (kod/nn/heads/yolov5.py:139-178): the box / objectness / class
1x1 convs of one pyramid level as ONE GEMM (N = A * (5 + nc)), parameters box_head.conv.*, obj_head.conv.*,
cls_head.conv.* with the reference's bias initialisation.

Yolov5BoxHead / Yolov5ObjectnessHead / Yolov5ClassificationHead (kod/nn/heads/yolov5.py:12-136): the three pieces as
modules of their own - one biased 1x1 conv (`conv.weight`, `conv.bias`) + the 'b (a p) h w -> b a h w p' view.  Each runs
the same fused head kernel (its GEMM is 48 columns wide whatever the head: the columns of the two absent pieces are zero
weights that are not parameters of the module) and returns its own slice."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

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


class _PieceHead(GraphModule):
    """One of the three head convs as a module: parameters conv.weight [A * P, C, 1, 1], conv.bias [A * P]."""
    _piece, _lo = "box", 0

    def _setup(self, in_channels: int, A: int, P: int, nc_graph: int, stride: int, shift: float):
        self.graph = build_head_graph(in_channels, A, nc_graph, stride)
        conv = nn.Conv2d(in_channels, A * P, kernel_size=1, stride=1)          # (the one RNG draw of the reference's constructor)
        if shift:
            with torch.no_grad():
                conv.bias.add_(shift)                                          # heads/yolov5.py:65-73,113-121
        self.conv = conv
        self._P = P
        # the two absent pieces of the fused GEMM: zero weights / biases, not parameters of this module
        self._absent = {}
        for key, p in (("box", 4), ("obj", 1), ("cls", nc_graph)):
            if key != self._piece:
                self._absent[f"{key}_head.conv.weight"] = nn.Parameter(torch.zeros(A * p, in_channels, 1, 1), requires_grad=False)
                self._absent[f"{key}_head.conv.bias"] = nn.Parameter(torch.zeros(A * p), requires_grad=False)
        self._engine = self._engine_device = None
        self._last_heads = ()
        self.engine_options = None

    def _engine_params(self):
        dev = self.conv.weight.device
        for k, v in self._absent.items():
            if v.device != dev:
                self._absent[k] = nn.Parameter(torch.zeros(v.shape, device=dev), requires_grad=False)
        return {f"{self._piece}_head.conv.weight": self.conv.weight, f"{self._piece}_head.conv.bias": self.conv.bias, **self._absent}

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        t = self._run([x])[0][0]                  # [B, A, h, w, 5 + nc_graph]
        return t[..., self._lo:self._lo + self._P]


def _prior(prior_probability: float) -> float:
    return -math.log((1 - prior_probability) / prior_probability)


class Yolov5BoxHead(_PieceHead):
    """kod.nn.heads.yolov5.Yolov5BoxHead (heads/yolov5.py:12-43): 4 A channels -> [B, A, h, w, 4]"""
    _piece, _lo = "box", 0

    def __init__(self, in_channels: int, num_anchors_per_cell: int):
        super().__init__()
        self._setup(in_channels, num_anchors_per_cell, 4, 1, 8, 0.0)


class Yolov5ClassificationHead(_PieceHead):
    """kod.nn.heads.yolov5.Yolov5ClassificationHead (heads/yolov5.py:46-91): nc A channels -> [B, A, h, w, nc]"""
    _piece, _lo = "cls", 5

    def __init__(self, in_channels: int, num_anchors_per_cell: int, num_classes: int, prior_probability: float = 0.01,
                 use_yv5_init: bool = True):
        super().__init__()
        shift = math.log(0.6 / (num_classes - 0.99999)) if use_yv5_init else _prior(prior_probability)
        self._setup(in_channels, num_anchors_per_cell, num_classes, num_classes, 8, shift)


class Yolov5ObjectnessHead(_PieceHead):
    """kod.nn.heads.yolov5.Yolov5ObjectnessHead (heads/yolov5.py:94-136): A channels -> [B, A, h, w, 1]"""
    _piece, _lo = "obj", 4

    def __init__(self, in_channels: int, num_anchors_per_cell: int, stride: int, prior_probability: float = 0.01,
                 use_yv5_init: bool = True):
        super().__init__()
        shift = math.log(8 / (640 / stride) ** 2) if use_yv5_init else _prior(prior_probability)
        self._setup(in_channels, num_anchors_per_cell, 1, 1, stride, shift)
