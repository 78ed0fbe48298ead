"""LR schedule functions (mirror kod/nn/optim/schedulers.py:13-24; the reference's scheduler classes only wrap
these in torch LambdaLR and cannot be constructed under torch >= 2.2 because of the removed `verbose` kwarg)."""
from __future__ import annotations

import math


def sch_cosine(x: int, max_epochs: int, lrf: float) -> float:
    return 1 + 0.5 * (lrf - 1) * (1 - math.cos((x / max_epochs) * math.pi))


def sch_linear(x: int, max_epochs: int, lrf: float) -> float:
    return (1 - x / max_epochs) * (1.0 - lrf) + lrf


def sch_cosine_annealing(x: int, max_epochs: int, lrf: float) -> float:
    return ((1 + math.cos(x * math.pi / max_epochs)) / 2) * (1 - lrf) + lrf
