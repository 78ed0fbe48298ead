"""Mirrors kod/core/types.py:6-8."""
from __future__ import annotations

from typing import NamedTuple


class FeatureShape(NamedTuple):
    width: int
    height: int
