"""bf16 storage emulation on top of the fp32 oracle network.

TEST INFRASTRUCTURE - see oracle/__init__.py.  The HIP path keeps weights (for the MFMA), pre-BN conv
outputs, activations and activation gradients in bf16 with fp32 accumulation.  ``emulate(net)`` inserts
straight-through bf16 rounding at exactly those points of the oracle so that the HIP kernels can be
checked TIGHTLY against it, while the un-emulated oracle (pinned to the reference) gives the stated
fp32-vs-bf16 tolerance.  Parameters keep their fp32 values (master weights); only the value used by the
convolution is rounded.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class _RoundBoth(torch.autograd.Function):
    """y = bf16(x) forward, g = bf16(g) backward."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


class _RoundFwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


rb = _RoundBoth.apply
rf = _RoundFwd.apply


def _conv_forward(self, x):
    y = F.conv2d(x, rf(self.weight), None if self.bias is None else self.bias, self.stride, self.padding)
    return rb(y) if self.bias is None else y        # head convs (bias) emit fp32


def emulate(net: nn.Module, image_too: bool = True) -> nn.Module:
    """Patch `net` (an OracleYolov5) in place."""
    for m in net.modules():
        if isinstance(m, nn.Conv2d):
            m.forward = _conv_forward.__get__(m, nn.Conv2d)
        elif isinstance(m, nn.SiLU):
            m.register_forward_hook(lambda mod, i, o: rb(o))
        elif isinstance(m, nn.MaxPool2d) or isinstance(m, nn.Upsample):
            m.register_forward_hook(lambda mod, i, o: rb(o))
    if image_too:
        net.register_forward_pre_hook(lambda mod, args: (rf(args[0]),))
    # residual adds: the HIP kernel rounds silu(bn(y)) + identity once; the hook above rounds the SiLU output
    # first, so block outputs can differ by one extra bf16 rounding (<= 2^-9 relative).
    from .network import Bottleneck
    for m in net.modules():
        if isinstance(m, Bottleneck) and m.identity:
            m.register_forward_hook(lambda mod, i, o: rb(o))
    return net
