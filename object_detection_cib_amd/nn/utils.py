"""Channel / depth rounding of the scaled YOLOv5 variants (kod/nn/utils.py:7-22): every width of the network is
`make_divisible(base, widen_factor)`, every CSP block count `make_round(base, deepen_factor)` - yv5n / s / m / l / x are
(0.25, 0.33) / (0.5, 0.33) / (0.75, 0.67) / (1, 1) / (1.25, 1.33).  The engine's static program (engine/graph.py) is built
from these two functions, so the channel counts the kernels see are the reference's by construction."""
import math


def make_divisible(x: float, widen_factor: float = 1.0, divisor: int = 8) -> int:
    """Smallest multiple of `divisor` that is >= x * widen_factor (kod/nn/utils.py:7-13).  The HIP kernels want channel
    counts in multiples of 8 (16-byte bf16 chunks): with the reference's divisor of 8 that holds for every variant."""
    return math.ceil(x * widen_factor / divisor) * divisor


def make_round(x: float, deepen_factor: float = 1.0) -> int:
    """Block count x * deepen_factor rounded half-to-even as Python's round does, never below 1; a count of 1 or less is
    passed through unscaled (kod/nn/utils.py:16-22)."""
    if x <= 1:
        return int(x)
    return int(max(round(x * deepen_factor), 1))
