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


class PeerExchange:
    """SyncBN statistic exchange over IPC-mapped peer buffers (csrc/comm.hip kodhip_peer_*, csrc/bn_act.hip *_peer).

    Replaces the per-layer all-reduces torch SyncBatchNorm issues under Lightning's `sync_batchnorm: True`
    (kod/configs/trainer/ddp.yaml:9): every rank of the node owns one exchange buffer, all ranks map all buffers, and the
    BatchNorm finalize / coefficient kernels read the other ranks' sums directly (one hop over xGMI, no collective
    launch).  The torch.distributed group only carries the 64-byte IPC handles at start-up.  `selftest()` compares the
    transport against the group's own all-reduce on random values before the engine trusts it."""

    def __init__(self, group, device, granules: int):
        """Collective over `group`.  Every rank takes part in the same two exchanges (IPC handles, then success flags)
        whatever happens locally, so a rank whose allocation / export / mapping fails makes EVERY rank raise the same
        RuntimeError instead of leaving the others inside a collective."""
        import torch.distributed as dist
        self.lib = _lib.lib()
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.device = device
        self.granules = int(granules)
        self._handle = None
        err, mine = None, None
        with torch.cuda.device(device):
            try:
                handle = C.c_void_p()
                _lib.check(self.lib.kodhip_peer_create(C.byref(handle), self.rank, self.world, self.granules), "peer_create")
                self._handle = handle
                buf = (C.c_ubyte * 64)()
                _lib.check(self.lib.kodhip_peer_export(handle, buf), "peer_export")
                mine = bytes(buf)
            except RuntimeError as e:
                err = str(e)
            everyone = [None] * self.world
            dist.all_gather_object(everyone, mine, group=group)                       # exchange 1: always
            if err is None and all(h is not None for h in everyone):
                try:
                    blob = (C.c_ubyte * (64 * self.world)).from_buffer_copy(b"".join(everyone))
                    _lib.check(self.lib.kodhip_peer_connect(self._handle, blob), "peer_connect")
                except RuntimeError as e:
                    err = str(e)
            elif err is None:
                err = "another rank could not create / export its exchange buffer"
            errs = [None] * self.world
            dist.all_gather_object(errs, err, group=group)                            # exchange 2: always
        if any(e is not None for e in errs):
            self.close()
            raise RuntimeError("peer exchange set-up failed: " + "; ".join(f"rank {r}: {e}" for r, e in enumerate(errs) if e))
        assert self.lib.kodhip_peer_view_bytes() == C.sizeof(_PeerView)
        self.view = _PeerView()
        _lib.check(self.lib.kodhip_peer_view(self._handle, C.byref(self.view)), "peer_view")
        dist.barrier(group=group)          # every rank has mapped every buffer before anyone publishes

    def view_ptr(self):
        return C.cast(C.byref(self.view), C.c_void_p)

    def step_begin(self, stream: int | None = None):
        self.check()
        s = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
        _lib.check(self.lib.kodhip_peer_step_begin(self._handle, s), "peer_step_begin")

    def all_reduce_f64(self, src: torch.Tensor, dst: torch.Tensor, slot: int = 0, stream: int | None = None):
        assert src.is_cuda and src.dtype == torch.float64 and dst.dtype == torch.float64 and src.numel() == dst.numel()
        s = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
        _lib.check(self.lib.kodhip_peer_allreduce_f64(self._handle, src.data_ptr(), dst.data_ptr(), src.numel(), slot, s),
                   "peer_allreduce_f64")

    def status(self) -> int:
        """0 ok, 1 a poll gave up (a peer never published), 2 the ranks' step counters diverged.  Reads the pinned host
        mirror of the verdict: no device synchronisation, cheap enough for once per step."""
        flag = C.c_int(0)
        _lib.check(self.lib.kodhip_peer_status(self._handle, C.byref(flag)), "peer_status")
        return int(flag.value)

    def check(self):
        """Raises when an exchange of an earlier step failed (the kernels already turned that step's statistics into NaN).
        Called by the engine before every training step and by GraphedTrainStep before every replay."""
        st = self.status() if self._handle else 0
        if st:
            raise RuntimeError("SyncBN peer exchange failed on rank %d: %s - the BatchNorm statistics of that step are NaN; "
                               "stop the job (KODHIP_SYNCBN=rccl selects the RCCL exchanges)" %
                               (self.rank, "a peer never published its sums (dead or stalled rank)" if st == 1 else
                                "the ranks' step counters diverged (uneven number of training forwards)"))

    def timed_out(self) -> bool:
        flag = C.c_int(0)
        _lib.check(self.lib.kodhip_peer_timed_out(self._handle, C.byref(flag)), "peer_timed_out")
        return bool(flag.value)

    def selftest(self, rounds: int = 3) -> bool:
        """The transport against the group's own all-reduce: `rounds` exchanges of rank-dependent random fp64 values over
        the whole buffer width.  True only when every rank saw exact agreement and no poll timed out (one verdict per
        round for the whole job: every rank stops together)."""
        import torch.distributed as dist
        n = min(self.granules // 2, 2048)
        for k in range(rounds):
            g = torch.Generator().manual_seed(1000 * k + self.rank)
            v = torch.randn(n, generator=g, dtype=torch.float64)
            src, dst = v.to(self.device), torch.empty(n, dtype=torch.float64, device=self.device)
            self.step_begin()
            self.all_reduce_f64(src, dst, 0)
            parts = [None] * self.world
            dist.all_gather_object(parts, v, group=self.group)
            want = parts[0].clone()
            for p in parts[1:]:
                want += p                          # rank order, like the kernel
            ok = torch.equal(dst.cpu(), want) and not self.timed_out()
            flags = [None] * self.world
            dist.all_gather_object(flags, bool(ok), group=self.group)
            if not all(flags):
                return False
        return True

    def close(self):
        if self._handle:
            torch.cuda.synchronize(self.device)
            _lib.check(self.lib.kodhip_peer_destroy(self._handle), "peer_destroy")
            self._handle = None


class _PeerView(C.Structure):
    _fields_ = [("peers", C.c_void_p * 8), ("world", C.c_int), ("rank", C.c_int), ("seq", C.c_void_p), ("timeout_flag", C.c_void_p), ("host_flag", C.c_void_p),
                ("max_spins", C.c_long)]
