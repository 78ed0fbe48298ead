"""Low-latency device -> host hand-off for the few places the validation path needs values on the host (NMS counts,
mAP match flags).  Two things make the obvious `.cpu()` / `.tolist()` expensive on this stack: a blocking copy parks
the host thread on an interrupt (millisecond wake-ups against ~10 ms of GPU work per validation batch), and every new
pinned allocation (what a pageable copy stages through, or a fresh `pin_memory()`) costs tens of milliseconds.  Here:
ONE pinned arena allocated once (grown only if a request does not fit), asynchronous copies into slices of it, one
event, and a polling wait on `event.query()`."""
from __future__ import annotations

from typing import List

import torch

_ARENA_BYTES = 8 << 20
_arena = None


def _get_arena(nbytes: int) -> torch.Tensor:
    global _arena
    if _arena is None or _arena.numel() < nbytes:
        _arena = torch.empty(max(_ARENA_BYTES, 2 * nbytes), dtype=torch.uint8).pin_memory()
    return _arena


def fetch(*tensors: torch.Tensor) -> List[torch.Tensor]:
    """Host copies of device tensors (same shapes / dtypes).  The returned tensors are views of the shared pinned
    arena: they are valid until the next fetch(), so consume them (or `.clone()` / `.numpy().copy()`) before that."""
    srcs = [t.contiguous() for t in tensors]
    offs, total = [], 0
    for t in srcs:
        total = (total + 63) // 64 * 64
        offs.append(total)
        total += t.numel() * t.element_size()
    arena = _get_arena(total + 64)
    outs = []
    for t, o in zip(srcs, offs):
        n = t.numel() * t.element_size()
        host = arena[o:o + n].view(t.dtype).view(t.shape)
        host.copy_(t, non_blocking=True)
        outs.append(host)
    ev = torch.cuda.Event()
    ev.record()
    while not ev.query():          # poll: no interrupt-driven sleep
        pass
    return outs
