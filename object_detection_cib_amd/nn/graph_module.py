"""GraphModule - an nn.Module whose forward / backward run a static sub-network program on the HIP engine.

The reference builds its network out of importable nn.Modules (kod/nn/layers/csp.py, sppf.py, backbones/yolov5.py,
necks/yolov5_pafpn.py, heads/yolov5.py); Yolov5Network here is ONE engine program, and these classes give the same
pieces the same names, constructor signatures, parameter paths (so `state_dict()` keys and seeded initial weights
equal the reference's) and call signatures - each as a small engine program of its own (engine/graph.py build_*_graph).
NCHW tensors in and out; the layout change at the module edge is a torch copy, every arithmetic op is a libkodhip
kernel.  There is no CPU path.
"""
from __future__ import annotations

import math
from typing import Sequence

import torch
import torch.nn as nn

from ..engine.executor import Engine, BN_EPS, BN_MOMENTUM
from ..engine.graph import Graph, head_param


class _Slot(nn.Module):
    """Name-space node: only holds children so parameter paths equal the reference's."""

    def forward(self, *a, **k):       # pragma: no cover
        raise RuntimeError("holder module: the sub-network runs through the HIP engine, not nn.Module.forward")


def ensure_path(root: nn.Module, path: str) -> nn.Module:
    node = root
    for part in path.split("."):
        if not part:
            continue
        if part not in node._modules:
            node.add_module(part, _Slot())
        node = node._modules[part]
    return node


def batchnorm_constants(norm_layer):
    """(eps, momentum) of the BatchNorm2d a `norm_layer` callable builds.  The HIP path implements train / eval BatchNorm2d
    with affine parameters and running statistics for any eps and any float momentum (the kernels take both as arguments;
    the reference's default is eps 1e-3, momentum .03, kod/nn/networks/yolov5.py:24); other normalisations are refused
    rather than silently replaced."""
    if norm_layer is None:
        return BN_EPS, BN_MOMENTUM
    probe = norm_layer(8)
    if not (isinstance(probe, nn.BatchNorm2d) and probe.affine and probe.track_running_stats and probe.momentum is not None):
        raise ValueError("the HIP path implements nn.BatchNorm2d (affine, running statistics, float momentum) only")
    return float(probe.eps), float(probe.momentum)


ACT_SILU, ACT_RELU, ACT_LEAKY, ACT_HARDSWISH, ACT_IDENTITY = 0, 1, 2, 3, 4      # csrc/bn_act.hip


def activation_code(activation_layer):
    """(code, negative slope) of the elementwise activation an `activation_layer` callable builds (torchvision's
    Conv2dNormActivation semantics: None = no activation).  SiLU - the reference's only configured choice - runs on the
    tuned kernels and fused epilogues; ReLU / LeakyReLU / Hardswish / identity on the plain passes (csrc/bn_act.hip
    bn_act_*); anything else is refused rather than silently replaced."""
    if activation_layer is None:
        return ACT_IDENTITY, 0.0
    probe = activation_layer()
    if isinstance(probe, nn.SiLU) or type(probe).__name__ in ("SiLU", "SiLUInplace"):
        return ACT_SILU, 0.0
    if isinstance(probe, nn.LeakyReLU):
        return ACT_LEAKY, float(probe.negative_slope)
    if isinstance(probe, nn.ReLU):
        return ACT_RELU, 0.0
    if isinstance(probe, nn.Hardswish):
        return ACT_HARDSWISH, 0.0
    if isinstance(probe, nn.Identity):
        return ACT_IDENTITY, 0.0
    raise ValueError("the HIP path implements SiLU, ReLU, LeakyReLU, Hardswish and Identity / None activations "
                     f"(got {type(probe).__name__})")


def check_norm_act(norm_layer, activation_layer):
    """BatchNorm2d (any eps / momentum, see batchnorm_constants) + one of the activations of activation_code(); anything
    else is refused rather than silently replaced.  Returns (activation code, slope)."""
    batchnorm_constants(norm_layer)
    return activation_code(activation_layer)


def add_unit_parameters(root: nn.Module, graph: Graph, norm_layer):
    """conv(bias=False) + BatchNorm holders in the graph's registration order (= the reference's construction order)."""
    for u in graph.units:
        slot = ensure_path(root, u.name)
        cin = 3 if u.stem else u.cin
        slot.add_module("0", nn.Conv2d(cin, u.cout, 6 if u.stem else u.k, u.s, u.p, bias=False))
        slot.add_module("1", norm_layer(u.cout))


def add_head_parameters(root: nn.Module, graph: Graph, use_yv5_init: bool = True, prior_probability: float = 0.01):
    """The three biased 1x1 convs of every head, with the reference's bias initialisation (heads/yolov5.py:65-73,113-121):
    use_yv5_init: objectness + log(8 / (640 / stride)^2), classes + log(0.6 / (nc - 0.99999)); otherwise both get the
    focal-loss prior - log((1 - p) / p) of `prior_probability`.  The box head keeps torch's default initialisation."""
    A, nc = graph.num_anchors, graph.num_classes
    prior = -math.log((1 - prior_probability) / prior_probability)
    for h in graph.heads:
        for key, p, shift in (("box", 4, 0.0), ("obj", 1, math.log(8 / (640 / h.stride) ** 2) if use_yv5_init else prior),
                              ("cls", nc, math.log(0.6 / (nc - 0.99999)) if use_yv5_init else prior)):
            conv = nn.Conv2d(h.cin, A * p, 1)
            if shift:
                with torch.no_grad():
                    conv.bias.add_(shift)
            ensure_path(root, head_param(h, key, "weight")[:-len(".conv.weight")]).add_module("conv", conv)


class _GraphFn(torch.autograd.Function):
    """The whole sub-network as one autograd node: forward / backward are the engine's op lists."""

    @staticmethod
    def forward(ctx, mod, n_heads, _anchor, *xs):
        ctx.mod, ctx.n_heads, ctx.n_in = mod, n_heads, len(xs)
        outs = mod._engine.forward(xs if mod.graph.inputs else xs[0], training=True)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        nh = ctx.n_heads
        eng = ctx.mod._engine
        head_grads = [g if g is not None else torch.zeros_like(o) for g, o in zip(grads[:nh], ctx.mod._last_heads)]
        gin = eng.backward(head_grads, list(grads[nh:]))
        if gin is None:
            gin = [None] * ctx.n_in
        return (None, None, None, *gin)


class GraphModule(nn.Module):
    """Base of the sub-network modules: parameters as torch holders, execution through an Engine over `self.graph`."""

    def _init_graph(self, graph: Graph, norm_layer, use_yv5_init: bool = True, prior_probability: float = 0.01,
                    activation=(ACT_SILU, 0.0)):
        from .networks.yolov5 import Yolov5BatchNorm2d
        self.graph = graph
        self._act = tuple(activation)           # (code, slope) from check_norm_act / activation_code
        self._bn_eps, self._bn_momentum = batchnorm_constants(norm_layer)
        add_unit_parameters(self, graph, norm_layer or Yolov5BatchNorm2d)
        if graph.heads:
            add_head_parameters(self, graph, use_yv5_init, prior_probability)
        self._engine = None
        self._engine_device = None
        self._last_heads = ()
        self.engine_options = None

    def engine(self) -> Engine:
        dev = next(self.parameters()).device
        if self._engine is None or self._engine_device != dev:
            if dev.type != "cuda":
                raise RuntimeError(f"{type(self).__name__} (HIP) must be on an MI355X: call .cuda() first; no CPU fallback")
            eng = Engine(self.graph, self._engine_params(), dict(self.named_buffers()), self.engine_options)
            eng.bn_eps, eng.bn_momentum = getattr(self, "_bn_eps", BN_EPS), getattr(self, "_bn_momentum", BN_MOMENTUM)
            eng.set_activation(*getattr(self, "_act", (ACT_SILU, 0.0)))
            eng._build_arenas(dev)
            self._engine, self._engine_device = eng, dev
        return self._engine

    def _engine_params(self):
        """engine parameter path -> tensor (the module's own parameters; a piece of a fused head adds the absent pieces)"""
        return dict(self.named_parameters())

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        if self._engine is not None:
            self._engine.mark_params_changed()
        return out

    def _run(self, xs: Sequence[torch.Tensor]):
        """-> (head tensors [B, A, h, w, 5+nc], output views NCHW fp32)"""
        eng = self.engine()
        xs = [x.float().contiguous() for x in xs]
        nh = len(self.graph.heads)
        if self.training and torch.is_grad_enabled():
            anchor = next(self.parameters())
            outs = _GraphFn.apply(self, nh, anchor, *xs)
            self._last_heads = tuple(outs[:nh])
        else:
            with torch.no_grad():
                outs = eng.forward(xs if self.graph.inputs else xs[0], training=self.training)
        return list(outs[:nh]), list(outs[nh:])
