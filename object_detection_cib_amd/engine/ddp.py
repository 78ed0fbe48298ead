"""Data-parallel helpers: gradient-bucket planning over the flat arena and rank sharding of sample indices.

Replaces what Lightning's `strategy: ddp` does implicitly for the reference
(kod/configs/trainer/ddp.yaml:4-9): torch DistributedDataParallel's reducer (bucketed all-reduce overlapped
with backward) and the DistributedSampler injection.  Pure host logic: testable on CPU with gloo.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch


def plan_buckets(unit_starts: Sequence[int], n_arena: int, bucket_elems: int) -> List[Tuple[int, int, int]]:
    """unit_starts[i] = arena offset where exec unit i's parameters begin (forward order, ascending).

    Gradients complete back-to-front.  Returns [(trigger_unit_index, lo, hi)]: once unit `trigger`'s wgrad has
    been enqueued, arena[lo:hi] is final and can be all-reduced.  Buckets tile [0, n_arena) exactly."""
    out = []
    hi = n_arena
    for i in range(len(unit_starts) - 1, -1, -1):
        lo = unit_starts[i]
        if hi - lo >= bucket_elems or i == 0:
            lo = 0 if i == 0 else lo
            out.append((i, lo, hi))
            hi = lo
    assert hi == 0
    return out


class _Joined:
    """Handle of a bucket all-reduce issued on a side stream: wait() makes the current stream depend on it."""

    def __init__(self, stream):
        self.stream = stream

    def wait(self):
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)


def launch_bucket(flat: torch.Tensor, lo: int, hi: int, group=None, stream=None, comm=None, also_after=None,
                  wait_caller: bool = True):
    """Sum all-reduce of a contiguous arena slice.

    GPU tensors: through the native RCCL communicator `comm` (engine/comm.py), or through torch.distributed when
    there is none (gloo-backed GPU tests).  With `stream` (a side stream) the collective overlaps the caller's
    stream: the side stream first waits for the caller's stream and for `also_after` (the weight-gradient stream
    that fills this bucket) - dependencies live in stream order only, so the code is hipGraph-capturable.  Without
    `stream` the collective is enqueued on the caller's stream itself: every collective of the step is then in ONE
    stream order, identical on all ranks - the conservative switch (KODHIP_COMM_OVERLAP=0); the engine's default passes
    the weight-gradient stream itself (Engine._comm_stream).
    wait_caller=False: `stream` already orders the collective behind everything that fills the bucket (the engine's
    weight-gradient stream: its last weight gradient waited for an event recorded after the bucket's BatchNorm / bias
    gradients) - no dependency on the caller's stream is added.  That matters under hipGraph capture: an edge from the
    main chain's latest kernel to the side stream, captured before the main chain's next kernel, makes the graph executor
    continue the MAIN chain on another queue and serialise the two branches (measured: -11 % step rate).
    CPU tensors (host-logic tests): a plain async_op Work."""
    if flat.is_cuda:
        cur = torch.cuda.current_stream()
        target = stream if stream is not None else cur
        if stream is not None and wait_caller:
            stream.wait_stream(cur)
        if also_after is not None and also_after is not target:      # e.g. the weight-gradient stream that fills this bucket
            target.wait_stream(also_after)
        if comm is not None:
            comm.all_reduce(flat[lo:hi], target.cuda_stream)
        else:
            with torch.cuda.stream(target):
                torch.distributed.all_reduce(flat[lo:hi], group=group)
        return _Joined(stream)
    return torch.distributed.all_reduce(flat[lo:hi], group=group, async_op=True)


def shard_indices(n: int, rank: int, world: int, seed: int = 0, epoch: int = 0, shuffle: bool = True,
                  drop_last: bool = False) -> List[int]:
    """torch.utils.data.DistributedSampler semantics (what Lightning injects): a seeded
    permutation per epoch, padded by wrap-around to a multiple of `world`, rank-strided."""
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        idx = torch.randperm(n, generator=g).tolist()
    else:
        idx = list(range(n))
    if drop_last and n % world:
        total = (n // world) * world
        idx = idx[:total]
    else:
        total = -(-n // world) * world
        pad = total - len(idx)
        if pad:
            idx += (idx * (pad // max(len(idx), 1) + 1))[:pad]
    return idx[rank:total:world]
