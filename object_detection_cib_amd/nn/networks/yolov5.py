"""Yolov5Network - drop-in for kod.nn.networks.yolov5.Yolov5Network (kod/nn/networks/yolov5.py:40-108).

Same constructor signature, same ``state_dict()`` (360 keys for yv5s: ``backbone.stem.0.weight`` ...
``hl_head.cls_head.conv.bias``), same seeded initial weights (parameters are created with the same
torch modules in the same order), same output structure - but ``forward`` runs the static HIP program of
``engine/`` instead of tracing nn.Modules, and the whole network is ONE autograd node whose backward is
the explicit HIP backward program.  There is no CPU path: forward raises off-GPU.
"""
from __future__ import annotations

import math
from functools import partial
from typing import Callable, NamedTuple

import os

import torch
import torch.nn as nn

from ...engine.graph import build_graph, P5_STAGES  # noqa: F401
from ...engine.executor import Engine, BN_EPS, BN_MOMENTUM
from ..heads.types import DetectionHeadResult
from ..layers.activations import SiLUInplace

Yolov5BatchNorm2d = partial(nn.BatchNorm2d, eps=BN_EPS, momentum=BN_MOMENTUM)     # networks/yolov5.py:24


class Yolov5NetworkResult(NamedTuple):
    ll: DetectionHeadResult
    ml: DetectionHeadResult
    hl: DetectionHeadResult


class _Slot(nn.Module):
    """Name-space node: only holds children so parameter paths equal the reference's."""

    def forward(self, *a, **k):       # pragma: no cover
        raise RuntimeError("holder module: the network runs through the HIP engine, not nn.Module.forward")


def _ensure(root: nn.Module, path: str) -> nn.Module:
    node = root
    for part in path.split("."):
        if part not in node._modules:
            node.add_module(part, _Slot())
        node = node._modules[part]
    return node


class _NetFn(torch.autograd.Function):
    """Whole-network autograd node: forward/backward are the engine's op lists."""

    @staticmethod
    def forward(ctx, net, x, _anchor):
        ctx.net = net
        outs = net._engine.forward(x, training=True)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        ctx.net._engine.backward(list(grads))
        return None, None, None


class Yolov5Network(nn.Module):
    def __init__(
        self,
        num_anchors_per_cell: int,
        num_classes: int,
        norm_layer: Callable[..., nn.Module] = Yolov5BatchNorm2d,
        activation_layer: Callable[..., nn.Module] = SiLUInplace,
        widen_factor: float = 1.0,
        deepen_factor: float = 1.0,
    ):
        super().__init__()
        from ..graph_module import batchnorm_constants, activation_code
        self._bn_eps, self._bn_momentum = batchnorm_constants(norm_layer)      # any BatchNorm2d eps / momentum; other normalisations are refused
        self._act = activation_code(activation_layer)      # SiLU (the reference's default) | ReLU | LeakyReLU | Hardswish | Identity / None
        self.num_classes = num_classes
        self.num_anchors_per_cell = num_anchors_per_cell
        self.widen_factor, self.deepen_factor = widen_factor, deepen_factor
        self.graph = build_graph(num_anchors_per_cell, num_classes, widen_factor, deepen_factor)
        # parameters: real torch modules as holders, created in the reference's construction order
        for u in self.graph.units:
            slot = _ensure(self, u.name)
            cin = 3 if u.stem else u.cin
            slot.add_module("0", nn.Conv2d(cin, u.cout, 6 if u.stem else u.k, u.s, u.p, bias=False))
            slot.add_module("1", norm_layer(u.cout))
        A, nc = num_anchors_per_cell, num_classes
        for h in self.graph.heads:
            for key, p, shift in (("box", 4, 0.0), ("obj", 1, math.log(8 / (640 / h.stride) ** 2)),
                                  ("cls", nc, math.log(0.6 / (nc - 0.99999)))):     # heads/yolov5.py:65-73,113-121
                conv = nn.Conv2d(h.cin, A * p, 1)
                if shift:
                    with torch.no_grad():
                        conv.bias.add_(shift)
                _ensure(self, f"{h.name}.{key}_head").add_module("conv", conv)
        self._engine: Engine | None = None
        self._engine_device = None
        self.engine_options = None         # an EngineOptions to build the engine with (None: EngineOptions.from_env())

    # ------------------------------------------------------------------ engine plumbing
    def engine(self) -> Engine:
        dev = next(self.parameters()).device
        if self._engine is None or self._engine_device != dev:
            if dev.type != "cuda":
                raise RuntimeError("Yolov5Network (HIP) must be on an MI355X: call .cuda() first; no CPU fallback")
            eng = Engine(self.graph, dict(self.named_parameters()), dict(self.named_buffers()), self.engine_options)
            eng.bn_eps, eng.bn_momentum = self._bn_eps, self._bn_momentum
            eng.set_activation(*self._act)
            eng._build_arenas(dev)
            self._engine, self._engine_device = eng, dev
        return self._engine

    def configure_distributed(self, process_group=None, sync_batchnorm: bool = True, bucket_mb: float = 8.0,
                              native_rccl: bool | None = None):
        """Data-parallel mode: RCCL all-reduce of gradient buckets overlapped with backward (+ SyncBN),
        the equivalent of Lightning's strategy=ddp / sync_batchnorm=True (kod/configs/trainer/ddp.yaml:4-9).

        native_rccl: True = collectives through this rank's own RCCL communicator (engine/comm.py; the group is
        only used to exchange the rendezvous id, so it may be gloo); None = that when the group is NCCL/RCCL-backed;
        False = torch.distributed calls on the group (gloo-backed tests with several ranks on one GPU)."""
        import torch.distributed as dist
        eng = self.engine()
        eng.process_group = process_group
        eng.world_size = dist.get_world_size(process_group)
        eng.sync_bn = sync_batchnorm
        # KODHIP_FORCE_COLLECTIVES=1 keeps the collective code path on a 1-rank group (single-GPU rehearsal of the N>1 path)
        eng.collectives = eng.world_size > 1 or eng.opt.force_collectives
        eng.bucket_bytes = int(bucket_mb * (1 << 20))
        eng.opt.bucket_mb = bucket_mb
        if native_rccl is None:
            native_rccl = dist.get_backend(process_group) == "nccl"
        if eng.collectives and native_rccl and eng.comm is None:
            from ...engine.comm import RcclComm
            eng.comm = RcclComm(process_group, eng.device)
            if eng.comm_overlap:          # default: gradient buckets on the weight-gradient stream, own communicator
                eng.comm_buckets = RcclComm(process_group, eng.device)
        if eng.collectives and sync_batchnorm and eng.peer is None and eng.opt.syncbn_exchange in ("auto", "peer"):
            self._setup_peer_exchange(eng, process_group)
        if eng.collectives:
            # rank 0's parameters and BatchNorm buffers everywhere (torch DDP does this at wrap time)
            for t in (eng.p_arena, eng.rm_arena, eng.rv_arena):
                if eng.comm is not None:
                    eng.comm.broadcast(t, 0)
                else:
                    dist.broadcast(t, src=dist.get_global_rank(process_group, 0) if process_group else 0,
                                   group=process_group)
            eng.mark_params_changed()

    @staticmethod
    def _setup_peer_exchange(eng, process_group):
        """SyncBN statistics over IPC-mapped peer buffers (engine/comm.py PeerExchange) when every rank sits on this
        node and the transport passes its start-up self-test; otherwise the exchanges stay RCCL all-reduces
        (KODHIP_SYNCBN=rccl forces that, =peer makes a failing set-up an error instead of a fallback)."""
        import socket
        import sys
        import torch.distributed as dist
        from ...engine.comm import PeerExchange
        strict = eng.opt.syncbn_exchange == "peer"
        hosts = [None] * eng.world_size
        dist.all_gather_object(hosts, socket.gethostname(), group=process_group)
        why = None
        if len(set(hosts)) != 1:
            why = "ranks on several nodes"
        elif eng.world_size > 8:
            why = "more than 8 ranks"
        if why is None:
            slots, off = {}, 0
            for u in eng.exec_units:
                for d in ("f", "b"):
                    slots[(u.name, d)] = off
                    off += 4 * u.cout
            try:
                peer = PeerExchange(process_group, eng.device, max(off, 4096))       # (raises on every rank or on none)
            except RuntimeError as e:          # no IPC between these processes (e.g. HSA_ENABLE_IPC_MODE_LEGACY unset)
                peer, why = None, str(e)
            if peer is not None:
                if peer.selftest():            # (one verdict for the whole job: the flags are gathered inside)
                    eng.peer, eng.peer_slots = peer, slots
                else:
                    peer.close()
                    why = "the transport self-test failed"
        if eng.peer is None:
            if strict:
                raise RuntimeError(f"KODHIP_SYNCBN=peer: {why}")
            if dist.get_rank(process_group) == 0:
                print(f"[kodhip] SyncBN statistics through RCCL all-reduces ({why})", file=sys.stderr, flush=True)
        eng.opt.syncbn_exchange = "peer" if eng.peer is not None else "rccl"

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        if self._engine is not None:
            self._engine.mark_params_changed()
        return out

    # ------------------------------------------------------------------ forward
    def forward_raw(self, x: torch.Tensor):
        """Three contiguous [B, A, h, w, 5+nc] fp32 tensors (ll, ml, hl)."""
        eng = self.engine()
        if x.dtype != torch.float32:
            x = x.float()
        x = x.contiguous()
        if self.training and torch.is_grad_enabled():
            anchor = next(self.parameters())
            return _NetFn.apply(self, x, anchor)
        with torch.no_grad():
            return tuple(eng.forward(x, training=self.training))

    def train_step(self, x: torch.Tensor, loss, image_feature_shape, targets, scale: float, image_ready: bool = False):
        """forward -> assigner + loss -> backward of `scale * (localization + classification + objectness)` (the
        reference's training_step, exp.py:104-121) as straight calls into the engine: no autograd graph, and the loss
        kernels run once (value and gradient together, `Yolov5Loss.value_and_grad`) instead of once per direction.
        Leaves the parameter gradients in `.grad` exactly like `total.backward()` does; returns (total, LossResult).
        The autograd route (`net(x)` -> `loss(...)` -> `.backward()`) stays available and gives the same numbers bit for
        bit; this is the route the captured step (engine/graphed.py) and bench.py take.
        image_ready: the batch already sits in the engine's input buffer (Engine.image_buffer), x carries only the shape."""
        eng = self.engine()
        assert self.training, "train_step() needs train mode"
        if x.dtype != torch.float32:
            x = x.float()
        with torch.no_grad():
            # the assignment depends only on the targets: its (three-block, latency-bound) kernel runs on a side stream
            # beside the forward pass instead of between the heads and the loss
            cur = torch.cuda.current_stream()
            # device-resident DetectionTargets (what a Lightning training_step passes) are concatenated by torch kernels
            # on the CURRENT stream: do that before the fork event, or the side stream's assignment kernel could read the
            # boxes before they are written
            from ...core.label_assignment.yv5 import BatchedTargets
            if not isinstance(targets, BatchedTargets):
                targets = BatchedTargets.from_targets(targets, eng.device)
            if eng.aux_stream is None:
                # the CSP branches' stream: side-stream work of one captured stream runs in launch order
                if eng.br_stream is None:
                    eng.br_stream = torch.cuda.Stream(device=eng.device)
                eng.aux_stream = eng.br_stream
            fork = torch.cuda.Event()
            fork.record(cur)
            asg = []

            def launch_assignment():
                # Launched - and captured - right after the forward chain's first layer, dependent only on the step's
                # start: the graph executor keeps a node's first captured successor on its queue (that must be the forward
                # chain) and runs the side streams' work in capture order on one queue (so this must come before the CSP
                # branches, or the loss would wait for it at the end of the forward pass).
                eng.aux_stream.wait_event(fork)
                asg.append(loss.assigner.assign_device(image_feature_shape, targets, eng.device, stream=eng.aux_stream))
            outs = eng.forward(x.contiguous(), training=True, after_first_layer=launch_assignment, image_ready=image_ready)
            cur.wait_stream(eng.aux_stream)
            asg = asg[0]
            lr, grads = loss.value_and_grad(image_feature_shape, outs, targets, (scale, scale, scale), assignment=asg)
            eng.backward(grads)
            total = loss._scaled_total          # scale * ((loc + cls) + obj), computed by the loss kernel in torch's order
            if total is None:
                total = scale * (lr.localization + lr.classification + lr.objectness)
        return total, lr

    def forward(self, x: torch.Tensor) -> Yolov5NetworkResult:
        raws = self.forward_raw(x)
        return Yolov5NetworkResult(*[DetectionHeadResult(t[..., 0:4], t[..., 4:5], t[..., 5:]) for t in raws])
