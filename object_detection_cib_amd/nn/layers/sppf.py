"""SPPFBottleneck - drop-in for kod.nn.layers.sppf.SPPFBottleneck (kod/nn/layers/sppf.py:14-84) on the HIP engine."""
from __future__ import annotations

from typing import Callable, Sequence

import torch
import torch.nn as nn

from ...engine.graph import build_sppf_graph
from .activations import SiLUInplace
from ..graph_module import GraphModule, check_norm_act


class SPPFBottleneck(GraphModule):
    """conv2(cat[x, pools...]), x = conv1(in) (or the input itself with use_conv_first=False); the pools write channel
    slices of the concat buffer.  kernel_sizes as in the reference (sppf.py:27-67): an int k = three cascaded k x k / stride 1
    pools (5 is what the network uses), a sequence = parallel pools of those sizes (run as a cascade where the sequence is
    one, e.g. the SPP windows (5, 9, 13)); any odd window up to 15.  Window 5 takes the tuned kernels
    (csrc/misc_ops.hip maxpool5_*), other windows the plain ones (maxpool_k_*)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_sizes: int | Sequence[int] = 5, use_conv_first: bool = True,
                 mid_channels_scale: float = 0.5, norm_layer: Callable[..., nn.Module] = nn.BatchNorm2d,
                 activation_layer: Callable[..., nn.Module] = SiLUInplace):
        super().__init__()
        act = check_norm_act(norm_layer, activation_layer)
        self.kernel_sizes = kernel_sizes
        self._init_graph(build_sppf_graph(in_channels, out_channels, mid_channels_scale, use_conv_first, kernel_sizes), norm_layer, activation=act)
        if not use_conv_first:
            self.conv1 = None                       # (the attribute exists and is None, sppf.py:39)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._run([x])[1][0]
