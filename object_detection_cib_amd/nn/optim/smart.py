"""SmartOptimizer - drop-in for kod.nn.optim.smart.SmartOptimizer (kod/nn/optim/smart.py:11-60).

Same constructor and call: ``SmartOptimizer(optimizer=partial(torch.optim.SGD, lr=.01, momentum=.937, nesterov=True),
weight_decay=5e-4)(net)`` (kod/configs/nn/optimizers/smart_sgd.yaml) returns a ``torch.optim.Optimizer`` whose
``param_groups`` are the reference's three named groups - bias_params | decay_params | norm_params, parameters in
module-walk order - so OptimizerWarmupUpdater (warmup.py:39-58), LambdaLR schedulers and Lightning checkpoints see
exactly what they see with the reference.  When ``net`` is the HIP ``Yolov5Network`` the returned optimizer is
``FusedSGD``: ``step()`` is ONE launch of ``sgd_nesterov_kernel`` (csrc/misc_ops.hip) over the engine's flat
parameter / gradient / momentum arenas instead of torch's foreach update over 189 tensors.
"""
from __future__ import annotations

from functools import partial
from typing import Callable

import torch
import torch.nn as nn


def split_param_groups(net: nn.Module):
    """smart.py:20-35: (bias, decay, no-decay norm weights) in module-walk order."""
    bias_params, no_decay_params, decay_params = [], [], []
    bn = tuple(v for k, v in nn.__dict__.items() if "Norm" in k and isinstance(v, type))
    for v in net.modules():
        for p_name, p in v.named_parameters(recurse=False):
            if p_name == "bias":
                bias_params.append(p)
            elif p_name == "weight" and isinstance(v, bn):
                no_decay_params.append(p)
            else:
                decay_params.append(p)
    return bias_params, decay_params, no_decay_params


class FusedSGD(torch.optim.Optimizer):
    """torch.optim.SGD (momentum with or without Nesterov, dampening, maximize) over a HIP Yolov5Network: same
    param_groups / state_dict layout, the update itself is the engine's fused multi-tensor kernel.  lr / momentum /
    weight_decay are read from ``param_groups`` at every ``step()`` (so warm-up and LR schedulers work unchanged);
    dampening, nesterov and maximize are the constructor's (one value for all groups, as SmartOptimizer builds them)."""

    def __init__(self, params, lr: float = 1e-3, momentum: float = 0.0, dampening: float = 0.0,
                 weight_decay: float = 0.0, nesterov: bool = False, *, net: nn.Module, world_size: int = 1,
                 maximize: bool = False, foreach=None, differentiable: bool = False):
        if nesterov and (momentum <= 0 or dampening != 0):            # torch.optim.SGD.__init__
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")
        if foreach is False or differentiable:
            raise NotImplementedError("FusedSGD is the fused multi-tensor update (foreach=False / differentiable=True have no HIP form)")
        defaults = dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov,
                        maximize=maximize, foreach=foreach, differentiable=differentiable)
        super().__init__(params, defaults)
        self.net = net
        self.world_size = world_size
        self._nesterov = bool(nesterov)
        self._dampening, self._maximize = float(dampening), bool(maximize)
        self.steps_taken = 0          # torch keeps no momentum_buffer before the first step (checkpoint layout)

    def _by_name(self):
        g = {pg.get("name"): pg for pg in self.param_groups}
        try:
            return [g["bias_params"], g["decay_params"], g["norm_params"]]
        except KeyError:
            raise RuntimeError("FusedSGD expects SmartOptimizer's groups bias_params / decay_params / norm_params")

    def hyper(self):
        eng = self.net.engine()                                # (the engine's device-side hyper block carries the flags)
        eng.sgd_nesterov, eng.sgd_dampening, eng.sgd_maximize = self._nesterov, self._dampening, self._maximize
        eng.sgd_steps = self.steps_taken
        g = self._by_name()
        return ([float(x["lr"]) for x in g], [float(x["momentum"]) for x in g], [float(x["weight_decay"]) for x in g])

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lr, mom, wd = self.hyper()
        self.net.engine().sgd_step(lr, mom, wd, 1.0 / self.world_size)
        self.steps_taken += 1
        return loss

    def state_dict(self):
        """torch.optim.SGD.state_dict() layout, momentum buffers read from the engine's arena (lightning/checkpoint.py)."""
        from ...lightning.checkpoint import optimizer_state_dict
        return optimizer_state_dict(self.net, self)

    def load_state_dict(self, sd):
        from ...lightning.checkpoint import load_optimizer_state_dict
        load_optimizer_state_dict(self.net, self, sd)


class SmartOptimizer(object):
    def __init__(self, optimizer: Callable[..., torch.optim.Optimizer], weight_decay: float):
        self.partial_opt = optimizer
        self.weight_decay = weight_decay

    def _factory(self, net: nn.Module):
        """torch.optim.SGD asked for a HIP network -> FusedSGD with the same keyword arguments."""
        from ..networks.yolov5 import Yolov5Network
        opt = self.partial_opt
        func = getattr(opt, "func", opt)
        if isinstance(net, Yolov5Network) and func in (torch.optim.SGD, FusedSGD):
            kw = dict(getattr(opt, "keywords", {}) or {})
            kw.pop("net", None)
            return partial(FusedSGD, *getattr(opt, "args", ()), **kw, net=net)
        return opt

    def __call__(self, net: nn.Module) -> torch.optim.Optimizer:
        bias_params, decay_params, no_decay_params = split_param_groups(net)
        optimizer = self._factory(net)(params=bias_params)                      # smart.py:36-58
        optimizer.param_groups[0]["name"] = "bias_params"
        optimizer.add_param_group(dict(params=decay_params, weight_decay=self.weight_decay, name="decay_params"))
        optimizer.add_param_group(dict(params=no_decay_params, weight_decay=0.0, name="norm_params"))
        return optimizer


def SmartSGD(net, lr: float = 0.01, momentum: float = 0.937, weight_decay: float = 5e-4, nesterov: bool = True,
             world_size: int = 1) -> FusedSGD:
    """The reference's configured optimizer (smart_sgd.yaml) in one call; `initial_lr` is pre-set like LambdaLR does."""
    opt = SmartOptimizer(partial(FusedSGD, lr=lr, momentum=momentum, nesterov=nesterov, net=net, world_size=world_size),
                         weight_decay)(net)
    for pg in opt.param_groups:
        pg.setdefault("initial_lr", pg["lr"])
    return opt
