"""CSPBlock / CSPLayer - drop-ins for kod.nn.layers.csp (kod/nn/layers/csp.py:16-111) on the HIP engine."""
from __future__ import annotations

from typing import Callable

import torch
import torch.nn as nn

from ...engine.graph import build_csp_block_graph, build_csp_layer_graph
from .activations import SiLUInplace
from ..graph_module import GraphModule, check_norm_act


class CSPBlock(GraphModule):
    """conv2_3x3(conv1_1x1(x)) [+ x when add_identity and in == out]; hidden = int(out * expand_ratio) (csp.py:16-58)."""

    def __init__(self, in_channels: int, out_channels: int, expand_ratio: float = 0.5, add_identity: bool = True,
                 norm_layer: Callable[..., nn.Module] = nn.BatchNorm2d, activation_layer: Callable[..., nn.Module] = SiLUInplace):
        super().__init__()
        act = check_norm_act(norm_layer, activation_layer)
        self.add_identity = add_identity and in_channels == out_channels
        self._init_graph(build_csp_block_graph(in_channels, out_channels, expand_ratio, add_identity), norm_layer, activation=act)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._run([x])[1][0]


class CSPLayer(GraphModule):
    """last_conv(cat[blocks(main_conv(x)), short_conv(x)]) with mid = int(out * expand_ratio) and `num_blocks` CSPBlocks of
    expand ratio 1 (csp.py:66-111); the concatenation is a channel-slice write, never a copy."""

    def __init__(self, in_channels: int, out_channels: int, expand_ratio: float = 0.5, add_identity: bool = True,
                 num_blocks: int = 1, norm_layer: Callable[..., nn.Module] = nn.BatchNorm2d,
                 activation_layer: Callable[..., nn.Module] = SiLUInplace):
        super().__init__()
        act = check_norm_act(norm_layer, activation_layer)
        self._init_graph(build_csp_layer_graph(in_channels, out_channels, expand_ratio, add_identity, num_blocks), norm_layer, activation=act)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._run([x])[1][0]
