"""Low-latency device -> host hand-off for the few places the validation path needs values on the host (NMS counts,
mAP match flags).  A blocking hipMemcpy / stream synchronize parks the host thread on an interrupt, whose wake-up
latency (milliseconds on this stack) dwarfs the ~10 ms of GPU work of a validation batch: measured 21-26 ms per batch
with `.tolist()` / `.cpu()` against 10 ms when nothing sleeps.  Here: asynchronous copies into cached pinned buffers,
one event, and a polling wait on `event.query()`."""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

_pinned: Dict[Tuple[torch.dtype, int], List[torch.Tensor]] = {}


def _buffer(dtype: torch.dtype, numel: int, slot: int) -> torch.Tensor:
    cap = 1 << max(int(numel - 1).bit_length(), 6)
    ring = _pinned.setdefault((dtype, cap), [])
    while len(ring) <= slot:
        ring.append(torch.empty(cap, dtype=dtype).pin_memory())
    return ring[slot]


def fetch(*tensors: torch.Tensor) -> List[torch.Tensor]:
    """Host copies of device tensors (same shapes / dtypes).  The returned tensors alias cached pinned buffers: they
    stay valid until the next fetch() that asks for a buffer of the same dtype and size class, so consume them (or
    `.clone()` / `.numpy().copy()`) before calling fetch again."""
    outs, used = [], {}
    for t in tensors:
        t = t.contiguous()
        key = (t.dtype, 1 << max(int(t.numel() - 1).bit_length(), 6))
        slot = used.get(key, 0)
        used[key] = slot + 1
        host = _buffer(t.dtype, max(t.numel(), 1), slot)[:t.numel()].view(t.shape)
        host.copy_(t, non_blocking=True)
        outs.append(host)
    ev = torch.cuda.Event()
    ev.record()
    while not ev.query():          # poll: no interrupt-driven sleep
        pass
    return outs
