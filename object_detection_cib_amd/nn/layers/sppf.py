"""SPPFBottleneck - drop-in for kod.nn.layers.sppf.SPPFBottleneck (kod/nn/layers/sppf.py:14-84) on the HIP engine."""
from __future__ import annotations

from typing import Callable, Sequence

import torch
import torch.nn as nn

from ...engine.graph import build_sppf_graph
from .activations import SiLUInplace
from ..graph_module import GraphModule, check_norm_act


def _as_cascade(kernel_sizes) -> bool:
    """True when the configuration is computed by three cascaded 5 x 5 / stride 1 max-pools: kernel_sizes = 5 (the
    cascade itself, sppf.py:49-55) or the parallel pools (5, 9, 13) (sppf.py:56-63) - a stride-1 max-pool of 5 applied
    j times IS the max-pool of 4 j + 1 (-inf padding), values and argmax routing alike."""
    if isinstance(kernel_sizes, int):
        return kernel_sizes == 5
    return tuple(kernel_sizes) == (5, 9, 13)


class SPPFBottleneck(GraphModule):
    """conv2(cat[x, p(x), p(p(x)), p(p(p(x)))]), x = conv1(in) (or the input itself with use_conv_first=False),
    p = MaxPool 5 / 1 / 2; the three pools write channel slices of the concat buffer.  kernel_sizes: 5 (what the network
    uses) or the SPP sequence (5, 9, 13), which is the same arithmetic; other kernel sizes are refused (the pool kernel
    is a 5 x 5 window, csrc/misc_ops.hip maxpool5_*)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_sizes: int | Sequence[int] = 5, use_conv_first: bool = True,
                 mid_channels_scale: float = 0.5, norm_layer: Callable[..., nn.Module] = nn.BatchNorm2d,
                 activation_layer: Callable[..., nn.Module] = SiLUInplace):
        super().__init__()
        check_norm_act(norm_layer, activation_layer)
        if not _as_cascade(kernel_sizes):
            raise NotImplementedError("the HIP SPPF implements kernel_sizes = 5 and the parallel form (5, 9, 13) "
                                      "(both are three cascaded 5 x 5 pools, sppf.py:49-63); other window sizes are not built")
        self.kernel_sizes = kernel_sizes
        self._init_graph(build_sppf_graph(in_channels, out_channels, mid_channels_scale, use_conv_first), norm_layer)
        if not use_conv_first:
            self.conv1 = None                       # (the attribute exists and is None, sppf.py:39)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._run([x])[1][0]
