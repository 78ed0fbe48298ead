"""Result tuple of one detection head (mirrors kod/nn/heads/types.py:8-11)."""
from __future__ import annotations

from typing import NamedTuple

import torch


class DetectionHeadResult(NamedTuple):
    box: torch.Tensor    # [B, A, h, w, 4]
    obj: torch.Tensor    # [B, A, h, w, 1]
    cls: torch.Tensor    # [B, A, h, w, nc]
