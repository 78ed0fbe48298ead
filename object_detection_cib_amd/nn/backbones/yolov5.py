"""StageConfig / Yolov5Backbone - drop-ins for kod.nn.backbones.yolov5 (kod/nn/backbones/yolov5.py:19-132)."""
from __future__ import annotations

from typing import Callable, NamedTuple, Sequence

import torch
import torch.nn as nn

from ...engine.graph import build_backbone_graph
from ..graph_module import GraphModule, check_norm_act


class StageConfig(NamedTuple):
    in_channels: int
    out_channels: int
    num_blocks: int
    add_identity: bool
    use_spp: bool


class Yolov5Backbone(GraphModule):
    """6x6 / s2 / p2 stem + stages (3x3 / s2 conv, CSPLayer, SPPF where use_spp); returns the list of stage outputs
    (backbones/yolov5.py:85-132).  Parameter paths stem.*, stages.stage<i>.blocks.<j>.* as in the reference."""

    def __init__(self, norm_layer: Callable[..., nn.Module], activation_layer: Callable[..., nn.Module], stages: list,
                 deepen_factor: float = 1.0, widen_factor: float = 1.0, spp_kernel_sizes: int | Sequence[int] = 5):
        super().__init__()
        act = check_norm_act(norm_layer, activation_layer)
        self._init_graph(build_backbone_graph([tuple(s) for s in stages], widen_factor, deepen_factor, spp_kernel_sizes), norm_layer, activation=act)

    def forward(self, x: torch.Tensor) -> list:
        return self._run([x])[1]
