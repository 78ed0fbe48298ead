"""RcclComm - this rank's RCCL communicator, driven through the C ABI (csrc/comm.hip).

The reference gets its collectives from torch DDP / SyncBatchNorm inside Lightning's ddp strategy
(kod/configs/trainer/ddp.yaml:4-9).  Here they are stream-ordered RCCL enqueues issued between the HIP kernels
of the step, so the whole step - collectives included - can be replayed as one hipGraph.  An existing
torch.distributed group is used once, to hand rank 0's rendezvous id to the other ranks.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from .. import _lib


class RcclComm:
    def __init__(self, group=None, device=None):
        import torch.distributed as dist
        self.lib = _lib.lib()
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        _lib.check(self.lib.kodhip_comm_load(bundled.encode() if os.path.exists(bundled) else None), "comm_load")
        ident = torch.zeros(128, dtype=torch.uint8)
        if self.rank == 0:
            _lib.check(self.lib.kodhip_comm_unique_id(ident.data_ptr()), "comm_unique_id")
        src = dist.get_global_rank(group, 0) if group is not None else 0
        carrier = ident.to(self.device) if dist.get_backend(group) == "nccl" else ident
        dist.broadcast(carrier, src=src, group=group)
        ident = carrier.cpu()
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.kodhip_comm_init(C.byref(handle), ident.data_ptr(), self.rank, self.world), "comm_init")
        self._handle = handle

    def all_reduce(self, t: torch.Tensor, stream: int | None = None):
        """In-place fp32 / fp64 sum over ranks, ordered on `stream` (a hipStream_t value; default: torch's current)."""
        assert t.is_cuda and t.dtype in (torch.float32, torch.float64) and t.is_contiguous() and self._handle
        s = torch.cuda.current_stream(t.device).cuda_stream if stream is None else stream
        _lib.check(self.lib.kodhip_comm_allreduce_sum(self._handle, t.data_ptr(), t.numel(), t.element_size(), s),
                   "comm_allreduce")

    def all_reduce_to(self, src: torch.Tensor, dst: torch.Tensor, stream: int | None = None):
        """dst = sum over ranks of src (src untouched)."""
        assert src.is_cuda and dst.is_cuda and src.dtype == dst.dtype and src.dtype in (torch.float32, torch.float64)
        assert src.is_contiguous() and dst.is_contiguous() and src.numel() == dst.numel() and self._handle
        s = torch.cuda.current_stream(src.device).cuda_stream if stream is None else stream
        _lib.check(self.lib.kodhip_comm_allreduce_sum_to(self._handle, src.data_ptr(), dst.data_ptr(), src.numel(),
                                                         src.element_size(), s), "comm_allreduce_to")

    def broadcast(self, t: torch.Tensor, root: int = 0, stream: int | None = None):
        assert t.is_cuda and t.is_contiguous() and self._handle
        s = torch.cuda.current_stream(t.device).cuda_stream if stream is None else stream
        _lib.check(self.lib.kodhip_comm_broadcast(self._handle, t.data_ptr(), t.numel() * t.element_size(), root, s),
                   "comm_broadcast")

    def group(self):
        """Context manager: the collectives issued inside are launched as ONE fused RCCL operation."""
        import contextlib

        @contextlib.contextmanager
        def _g():
            _lib.check(self.lib.kodhip_comm_group_start(), "comm_group_start")
            try:
                yield
            finally:
                _lib.check(self.lib.kodhip_comm_group_end(), "comm_group_end")
        return _g()

    def close(self):
        if self._handle:
            torch.cuda.synchronize(self.device)
            _lib.check(self.lib.kodhip_comm_destroy(self._handle), "comm_destroy")
            self._handle = None
