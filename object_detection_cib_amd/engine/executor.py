"""Executes the static YOLOv5 program (engine/graph.py) with libkodhip kernels.

Replaces what autograd + aten do for the reference's training step
(kod/lightning/experiments/yv5_baseline/exp.py:104-138 -> net forward, loss.backward): forward and
backward are explicit op lists over pre-allocated channels-last bf16 buffers; parameters, gradients and
momentum live in flat fp32 arenas (one fused SGD launch, contiguous all-reduce buckets); weights are
re-packed to bf16 MFMA layouts once per step.

PyTorch is used for device memory, streams and torch.distributed only.
"""
from __future__ import annotations

import os

from typing import Dict, Optional


import torch

from .. import _lib
from .graph import Graph
from .options import EngineOptions
from .arenas import ArenaMixin, _UnitState, _pad          # noqa: F401
from .buffers import BufferMixin
from .forward import ForwardMixin, BN_EPS, BN_MOMENTUM   # noqa: F401
from .backward import BackwardMixin


class Engine(ArenaMixin, BufferMixin, ForwardMixin, BackwardMixin):
    """Owns arenas + buffers for one network instance (one process, one GPU).  The work is split by concern:
    engine/options.py (every switch), engine/plan.py + engine/ddp.py (pure planning, CPU-testable), engine/arenas.py
    (parameters, packs, optimizer), engine/buffers.py (per-shape buffer sets), engine/forward.py / engine/backward.py
    (the launch programs), engine/comm.py (RCCL communicator, SyncBN peer exchange)."""

    def __init__(self, graph: Graph, params: Dict[str, torch.nn.Parameter], buffers: Dict[str, torch.Tensor],
                 options: Optional[EngineOptions] = None):
        _lib.require_gpu()
        self.lib = _lib.lib()
        self.opt = options or EngineOptions.from_env()      # every switch, read once (engine/options.py)
        self.g = graph
        self.params = params          # full state_dict-style names -> Parameter (shared with the nn.Module)
        self.buffers = buffers        # running_mean / running_var / num_batches_tracked
        self.device = None
        self.shape = None             # (B, H, W) of the current allocation
        # one complete buffer set per (B, H, W): a captured hipGraph bakes buffer addresses in, so a forward at another
        # shape (validation batch, partial last batch) must never free what a graph replays into
        self._sets: Dict[tuple, dict] = {}
        self._pinned = set()          # shapes a captured graph depends on: never evicted
        self.max_shape_sets = self.opt.max_shape_sets
        self.training_ready = False
        self.bn_eps, self.bn_momentum = BN_EPS, BN_MOMENTUM     # the network's BatchNorm2d constants (set by the nn.Module that owns the engine)
        self.act_kind, self.act_slope = 0, 0.0                  # the units' activation (csrc/bn_act.hip ACT_*; 0 = SiLU), see set_activation
        self.sync_bn = False
        self.process_group = None
        self.world_size = 1
        self.wg_stream = None          # side stream of the weight-gradient kernels (see backward)
        self.wg_more = []              # further weight-gradient streams (EngineOptions.wgrad_streams)
        self.wgrad_overlap = self.opt.wgrad_overlap
        # regions of the slab scratch = weight-gradient streams (the per-bucket reduction keeps a region per layer: one stream)
        self._wg_streams = 1 if self.opt.wgrad_reduce_batched else max(1, int(self.opt.wgrad_streams))
        self._wg_regions = self._wg_streams
        self.wgrad_fork = self.opt.wgrad_fork     # see backward(): deferred capture of the wgrad launches
        self.comm = None               # RcclComm when the process group is RCCL-backed (the product path)
        self.comm_buckets = None       # second communicator: gradient buckets on the weight-gradient stream (comm_overlap)
        self.comm_overlap = self.opt.comm_overlap
        self.peer = None               # PeerExchange: SyncBN sums over IPC-mapped peer buffers instead of RCCL all-reduces
        self.peer_slots = None         # {(unit name, "f" | "b"): first granule of that exchange}
        self.collectives = False       # True when gradients / BN sums go through the process group (world > 1)
        self.bucket_bytes = int(self.opt.bucket_mb * (1 << 20))
        self._checked_shapes = set()   # local batch shapes already compared across the ranks (SyncBN, see forward)
        self._pending = []
        self._packed_version = -1
        self.param_version = 0
        self.stats_version = 0         # bumped by every eager training forward (BatchNorm running statistics moved)
        self._eval_aff = None          # eval-mode BN constants of all units (flat), see _eval_affine_ptrs
        self._fork_ev = None
        self.br_stream = None         # side stream of the CSP short_conv branch in forward()
        self.head_stream = None       # side stream of the P3 / P4 head convolutions in forward()
        self.aux_stream = None        # side stream of work that only depends on the step's inputs (label assignment)
        self.branch_overlap = self.opt.branch_overlap
        self.profile = None           # list of (family, start_event, end_event, algorithmic bytes), see _t0 / _t1
        # KODHIP_DEBUG_STAMPS=1: device clock stamps at named points of the step, also inside a replayed hipGraph
        # (bench.py stamp_report): stamp_names[i] <-> stamp_buf[i]
        self.stamp_buf, self.stamp_names = None, []
        self.stamps_on = os.environ.get("KODHIP_DEBUG_STAMPS", "0") == "1"

    def set_activation(self, kind: int, slope: float = 0.0):
        """The conv units' activation (nn/graph_module.activation_code), to be set before the arenas are built.  Anything but
        SiLU runs on the plain elementwise passes (kodhip_bn_act_*): the fused forms that carry SiLU's arithmetic - the
        BatchNorm-backward reduction in the data gradients' epilogue, the fused stem backward, the pair apply - are off."""
        import dataclasses
        self.act_kind, self.act_slope = int(kind), float(slope)
        if self.act_kind != 0:
            self.opt = dataclasses.replace(self.opt, bn_reduce_fused=False, stem_bwd_fused=False, pair_fwd=0)
