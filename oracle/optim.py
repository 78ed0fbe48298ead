"""Oracle: SmartOptimizer grouping, Nesterov SGD step, warm-up, LR schedule.

TEST INFRASTRUCTURE - see oracle/__init__.py.  Restates:

* param groups           kod/nn/optim/smart.py:20-60
* SGD hyper-parameters   kod/configs/nn/optimizers/smart_sgd.yaml:1-8
* warm-up                kod/lightning/experiments/yv5_baseline/warmup.py:24-58,
                         trigger kod/lightning/experiments/yv5_baseline/exp.py:164-185
* linear LR schedule     kod/nn/optim/schedulers.py:19-20
* SGD update             torch.optim.SGD (momentum, nesterov, dampening 0)
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

LR0, MOMENTUM, WEIGHT_DECAY, LRF = 0.01, 0.937, 5e-4, 0.01
WARMUP_EPOCHS, WARMUP_BIAS_LR, WARMUP_MOMENTUM = 3, 0.1, 0.8
GROUP_NAMES = ("bias_params", "decay_params", "norm_params")


def param_groups(net: nn.Module):
    """smart.py:20-35 -> (bias, decay, norm) lists of parameters, module-walk order."""
    bias, decay, norm = [], [], []
    norm_types = tuple(v for k, v in nn.__dict__.items() if "Norm" in k and isinstance(v, type))
    for m in net.modules():
        for name, p in m.named_parameters(recurse=False):
            if name == "bias":
                bias.append(p)
            elif name == "weight" and isinstance(m, norm_types):
                norm.append(p)
            else:
                decay.append(p)
    return bias, decay, norm


def sch_linear(epoch: int, max_epochs: int = 300, lrf: float = LRF) -> float:
    return (1 - epoch / max_epochs) * (1.0 - lrf) + lrf


def warmup_values(step: int, epoch: int, nw: int, initial_lr: float = LR0, max_epochs: int = 300):
    """Per-group (lr, momentum) after OptimizerWarmupUpdater.__call__ (warmup.py:39-58)."""
    out = {}
    for g in GROUP_NAMES:
        lr = float(np.interp(step, [0, nw], [WARMUP_BIAS_LR if g == "bias_params" else 0.0,
                                             initial_lr * sch_linear(epoch, max_epochs)]))
        mom = float(np.interp(step, [0, nw], [WARMUP_MOMENTUM, MOMENTUM]))
        out[g] = (lr, mom)
    return out


def warmup_steps(batches_per_epoch: int) -> int:
    """exp.py:168-174."""
    return max(round(batches_per_epoch * WARMUP_EPOCHS), 100)


@torch.no_grad()
def sgd_nesterov_step(p: torch.Tensor, g: torch.Tensor, buf: torch.Tensor | None,
                      lr: float, momentum: float, wd: float):
    """One torch.optim.SGD(nesterov=True, dampening=0) update; returns the new buffer."""
    if wd:
        g = g + wd * p
    buf = g.clone() if buf is None else buf.mul_(momentum).add_(g)
    p.sub_(lr * (g + momentum * buf))
    return buf
