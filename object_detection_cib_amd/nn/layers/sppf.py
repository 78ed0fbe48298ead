"""SPPFBottleneck - drop-in for kod.nn.layers.sppf.SPPFBottleneck (kod/nn/layers/sppf.py:14-84) on the HIP engine."""
from __future__ import annotations

from typing import Callable, Sequence

import torch
import torch.nn as nn

from ...engine.graph import build_sppf_graph
from ..graph_module import GraphModule, check_norm_act


class SPPFBottleneck(GraphModule):
    """conv2(cat[x, p(x), p(p(x)), p(p(p(x)))]), x = conv1(in), p = MaxPool 5 / 1 / 2: the configuration the network uses
    (kernel_sizes = 5, use_conv_first = True); the three pools write channel slices of the concat buffer."""

    def __init__(self, in_channels: int, out_channels: int, kernel_sizes: int | Sequence[int] = 5, use_conv_first: bool = True,
                 mid_channels_scale: float = 0.5, norm_layer: Callable[..., nn.Module] = None,
                 activation_layer: Callable[..., nn.Module] = None):
        super().__init__()
        check_norm_act(norm_layer, activation_layer)
        if kernel_sizes != 5 or not use_conv_first:
            raise NotImplementedError("the HIP SPPF implements kernel_sizes=5 with the leading 1x1 conv (sppf.py:29-36,61-67)")
        self.kernel_sizes = kernel_sizes
        self._init_graph(build_sppf_graph(in_channels, out_channels, mid_channels_scale), norm_layer)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._run([x])[1][0]
