"""LR schedules - drop-ins for kod.nn.optim.schedulers (kod/nn/optim/schedulers.py:13-117).

Same classes, constructor signatures and the `sch_fn` attribute the warm-up hook reads
(kod/lightning/experiments/yv5_baseline/exp.py:178-184).  The reference forwards `verbose=` to torch's LambdaLR /
StepLR, a keyword torch >= 2.2 deprecated and 2.10 removed, so its classes cannot be constructed on this stack; here
`verbose` is accepted and ignored.  Quirks kept: CosineScheduler's `sch_fn` (warm-up target) is the LINEAR schedule
while its LambdaLR follows the cosine (schedulers.py:55-70); StepScheduler's `sch_fn` uses lrf = 0.1."""
from __future__ import annotations

import math
from functools import partial

from torch.optim.lr_scheduler import CosineAnnealingLR, LambdaLR, StepLR
from torch.optim.optimizer import Optimizer


def sch_cosine(x: int, max_epochs: int, lrf: float) -> float:
    return 1 + 0.5 * (lrf - 1) * (1 - math.cos((x / max_epochs) * math.pi))


def sch_linear(x: int, max_epochs: int, lrf: float) -> float:
    return (1 - x / max_epochs) * (1.0 - lrf) + lrf


def sch_cosine_annealing(x: int, max_epochs: int, lrf: float) -> float:
    return ((1 + math.cos(x * math.pi / max_epochs)) / 2) * (1 - lrf) + lrf


class LinearScheduler(LambdaLR):
    def __init__(self, lrf: float, optimizer: Optimizer, max_epochs: int, verbose: bool = False) -> None:
        self.sch_fn = partial(sch_linear, max_epochs=max_epochs, lrf=lrf)
        super().__init__(optimizer, lr_lambda=self.sch_fn)


class CosineScheduler(LambdaLR):
    def __init__(self, lrf: float, optimizer: Optimizer, max_epochs: int, verbose: bool = False) -> None:
        self.sch_fn = partial(sch_linear, max_epochs=max_epochs, lrf=lrf)
        super().__init__(optimizer, lr_lambda=partial(sch_cosine, max_epochs=max_epochs, lrf=lrf))


class StepScheduler(StepLR):
    def __init__(self, optimizer: Optimizer, step_size: int = 100, gamma: float = 0.5, last_epoch: int = -1,
                 max_epochs: int = -1, verbose: bool = False) -> None:
        self.sch_fn = partial(sch_linear, max_epochs=max_epochs, lrf=0.1)
        super().__init__(optimizer, step_size=step_size, gamma=gamma)


class CosineAnnealingScheduler(CosineAnnealingLR):
    def __init__(self, lrf: float, optimizer: Optimizer, max_epochs: int, verbose: bool = False) -> None:
        self.sch_fn = partial(sch_cosine_annealing, max_epochs=max_epochs, lrf=lrf)
        super().__init__(optimizer, T_max=max_epochs, last_epoch=-1)
