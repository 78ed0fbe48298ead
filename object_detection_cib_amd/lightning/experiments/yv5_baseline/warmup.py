"""OptimizerWarmupUpdater (mirrors kod/lightning/experiments/yv5_baseline/warmup.py:11-58; host scalar math)."""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np


class OptimizerWarmupUpdater(object):
    def __init__(self, warmup_epochs: int, warmup_bias_lr: float, warmup_momentum: float,
                 momentum: Optional[float] = None):
        self.warmup_bias_lr = warmup_bias_lr
        self.warmup_momentum = warmup_momentum
        self.warmup_epochs = warmup_epochs
        self.momentum = momentum

    def __call__(self, current_step: int, current_epoch: int, max_warmup_steps: int, sch_fn: Callable, optimizer):
        xi = [0, max_warmup_steps]
        for pg in optimizer.param_groups:
            pg["lr"] = np.interp(current_step, xi, [self.warmup_bias_lr if pg["name"] == "bias_params" else 0.0,
                                                    pg["initial_lr"] * sch_fn(current_epoch)])
            if "momentum" in pg:
                assert self.momentum is not None
                pg["momentum"] = np.interp(current_step, xi, [self.warmup_momentum, self.momentum])
