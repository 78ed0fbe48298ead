"""Yolov5PAFPN - drop-in for kod.nn.necks.yolov5_pafpn.Yolov5PAFPN (kod/nn/necks/yolov5_pafpn.py:16-202)."""
from __future__ import annotations

from typing import Callable, Sequence

import torch
import torch.nn as nn

from ...engine.graph import build_pafpn_graph
from ..graph_module import GraphModule, check_norm_act


class Yolov5PAFPN(GraphModule):
    """reduce(P5) -> top-down x2 (nearest upsample, concat, CSPLayer [+ 1x1 reduce]) -> bottom-up x2 (3x3 / s2, concat,
    CSPLayer); three pyramid levels at strides 8 / 16 / 32.  Upsamples and concats are channel-slice writes."""

    def __init__(self, in_channels_list: Sequence[int], norm_layer: Callable[..., nn.Module],
                 activation_layer: Callable[..., nn.Module], num_blocks: int = 3, expand_ratio: float = 0.5,
                 deepen_factor: float = 1.0, widen_factor: float = 1.0):
        super().__init__()
        act = check_norm_act(norm_layer, activation_layer)
        self.in_channels_list = in_channels_list
        self.widen_factor, self.deepen_factor, self.num_blocks = widen_factor, deepen_factor, num_blocks
        self._init_graph(build_pafpn_graph(in_channels_list, num_blocks, expand_ratio, deepen_factor, widen_factor), norm_layer, activation=act)

    def forward(self, inputs: Sequence[torch.Tensor]) -> tuple:
        assert len(inputs) == len(self.in_channels_list)
        return tuple(self._run(list(inputs))[1])
