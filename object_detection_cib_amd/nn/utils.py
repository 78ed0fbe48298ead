"""Channel / depth rounding (mirrors kod/nn/utils.py:7-22)."""
from ..engine.graph import make_divisible, make_round  # noqa: F401
