"""Executes the static YOLOv5 program (engine/graph.py) with libkodhip kernels.

Replaces what autograd + aten do for the reference's training step
(kod/lightning/experiments/yv5_baseline/exp.py:104-138 -> net forward, loss.backward): forward and
backward are explicit op lists over pre-allocated channels-last bf16 buffers; parameters, gradients and
momentum live in flat fp32 arenas (one fused SGD launch, contiguous all-reduce buckets); weights are
re-packed to bf16 MFMA layouts once per step.

PyTorch is used for device memory, streams and torch.distributed only.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import numpy as np
import os

import torch

from .. import _lib
from .graph import Graph, ConvUnit, HeadUnit, View, Buf
from .ddp import plan_buckets, launch_bucket
from .options import EngineOptions
from .plan import backward_writes, plan_f32_accumulation

BN_EPS, BN_MOMENTUM = 1e-3, 0.03        # kod/nn/networks/yolov5.py:24


def _pad(n: int, a: int = 64) -> int:
    return (n + a - 1) // a * a


class _UnitState:
    __slots__ = ("u", "w_off", "g_off", "b_off", "f_off", "d_off", "Kp", "Kdp", "rs_off", "stats", "T",
                 "sums", "aff", "bsums", "bsums_g", "bpart", "T2", "coef", "raw", "M", "H", "W", "Ho", "Wo",
                 "fused_red", "segs", "seg_slots", "Kp_f", "raw_ld", "s2_fold", "wg_splits", "wg_off")


class Engine:
    """Owns arenas + buffers for one network instance (one process, one GPU)."""

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
        self.sync_bn = False
        self.process_group = None
        self.world_size = 1
        self.wg_stream = None          # side stream of the weight-gradient kernels (see backward)
        self.wgrad_overlap = self.opt.wgrad_overlap
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

    # ------------------------------------------------------------------ arenas
    def _build_arenas(self, device):
        g = self.g
        order = []                    # (param name, group) in forward execution order, packed sets
        self.ustate: Dict[str, _UnitState] = {}
        off = 0
        layout = {}                   # name -> (offset, numel)
        gid = []

        def place(names, group):
            nonlocal off
            start = off
            for n in names:
                p = self.params[n]
                layout[n] = (off, p.numel())
                off += p.numel()
            end = _pad(off)
            gid.extend([group] * ((end - start) // 64))
            off = end

        exec_units = [op.unit for op in g.ops if op.kind == "conv"]
        for u in exec_units:
            place([u.name + ".0.weight"], 1)
            place([u.name + ".1.weight"], 2)
            place([u.name + ".1.bias"], 0)
        for h in g.heads:
            place([f"{h.name}.{k}_head.conv.weight" for k in ("box", "obj", "cls")], 1)
            place([f"{h.name}.{k}_head.conv.bias" for k in ("box", "obj", "cls")], 0)
        self.n_arena = off
        self.layout = layout
        self.unit_starts = ([layout[u.name + '.0.weight'][0] for u in exec_units]
                            + [layout[f'{h.name}.box_head.conv.weight'][0] for h in g.heads])
        self.p_arena = torch.zeros(off, dtype=torch.float32, device=device)
        self.g_arena = [torch.zeros(off, dtype=torch.float32, device=device) for _ in range(2)]
        self.g_cur = 0
        self.m_arena = torch.zeros(off, dtype=torch.float32, device=device)
        self.gid = torch.tensor(gid, dtype=torch.uint8, device=device)
        with torch.no_grad():
            for n, (o, k) in layout.items():
                p = self.params[n]
                self.p_arena[o:o + k].copy_(p.detach().reshape(-1).to(device))
                p.data = self.p_arena[o:o + k].view(p.shape)
                p.grad = None
        # BN running statistics arena
        roff = 0
        self.rs_layout = {}
        for u in exec_units:
            self.rs_layout[u.name] = roff
            roff += _pad(u.cout, 16)
        self.rm_arena = torch.zeros(roff, dtype=torch.float32, device=device)
        self.rv_arena = torch.ones(roff, dtype=torch.float32, device=device)
        self.nbt_arena = torch.zeros(len(exec_units), dtype=torch.int64, device=device)
        with torch.no_grad():
            for i, u in enumerate(exec_units):
                o = self.rs_layout[u.name]
                for key, arena in (("running_mean", self.rm_arena), ("running_var", self.rv_arena)):
                    b = self.buffers[f"{u.name}.1.{key}"]
                    arena[o:o + u.cout].copy_(b.to(device))
                    b.data = arena[o:o + u.cout]
                b = self.buffers[f"{u.name}.1.num_batches_tracked"]
                self.nbt_arena[i] = b.to(device)
                b.data = self.nbt_arena[i]
        # weight packs
        descs = []
        foff = doff = 0
        blk = 0
        A, nc = g.num_anchors, g.num_classes

        def add_desc(w_name, f_off, d_off, N, Cin, KH, KW, Kp, Kdp, Ntot, n_off, stem):
            nonlocal blk
            descs.append([layout[w_name][0], f_off, d_off, N, Cin, KH, KW, Kp, Kdp, Ntot, n_off, stem, blk])
            blk += (N * Cin * KH * KW + 255) // 256

        for u in exec_units:
            st = _UnitState()
            st.u = u
            K = u.k * u.k * u.cin if not u.stem else 144
            st.Kp = _pad(K, 32)                       # K of the weight-gradient slabs (stem: 6x3 taps x 8 = 144 -> 160)
            # forward operand rows: the stem packs each kernel row as one 32-value K step (4 pixel pairs, the 4th
            # zero) so that it runs on the LDS-DMA path like every other layer
            # packed MFMA operands: K axis tap-major, every tap padded to a multiple of 32 channels (csrc/misc_ops.hip)
            st.Kp_f = 6 * 32 if u.stem else u.k * u.k * _pad(u.cin, 32)
            st.Kdp = u.k * u.k * _pad(u.cout, 32)
            st.f_off, st.d_off = foff, (-1 if u.stem else doff)
            foff += u.cout * st.Kp_f
            s2 = (not u.stem) and u.k == 3 and u.s == 2 and u.p == 1
            st.s2_fold = bool(s2 and self.lib.kodhip_conv_dgrad_s2_folded(u.cin, u.cout))
            if s2:       # parity-class packs (1 + 2 + 2 + 4 taps) or the folded pack (4 classes x 4 taps), csrc/conv_igemm.hip
                doff += u.cin * (16 if st.s2_fold else 9) * _pad(u.cout, 32)
            elif not u.stem:
                doff += u.cin * st.Kdp
            st.w_off = layout[u.name + ".0.weight"][0]
            st.g_off = layout[u.name + ".1.weight"][0]
            st.b_off = layout[u.name + ".1.bias"][0]
            st.rs_off = self.rs_layout[u.name]
            if u.stem:
                add_desc(u.name + ".0.weight", st.f_off, -1, u.cout, 3, 6, 6, st.Kp_f, 0, 0, 0, 1)
            else:
                add_desc(u.name + ".0.weight", st.f_off, st.d_off, u.cout, u.cin, u.k, u.k, st.Kp_f, st.Kdp,
                         u.cout, 0, (3 if st.s2_fold else 2) if s2 else 0)
            self.ustate[u.name] = st
        self.hstate = {}
        self.head_npad = _pad(A * (5 + nc), 8)
        for h in g.heads:
            Kp = _pad(h.cin, 32)
            Kdp = _pad(self.head_npad, 32)
            hs = dict(f_off=foff, d_off=doff, Kp=Kp, Kdp=Kdp,
                      w_off=layout[f"{h.name}.box_head.conv.weight"][0],
                      b_off=layout[f"{h.name}.box_head.conv.bias"][0])
            n_off = 0
            for k, n in (("box", 4 * A), ("obj", A), ("cls", nc * A)):
                add_desc(f"{h.name}.{k}_head.conv.weight", foff + n_off * Kp, doff, n, h.cin, 1, 1, Kp, Kdp,
                         self.head_npad, n_off, 0)
                n_off += n
            foff += self.head_npad * Kp
            doff += h.cin * Kdp
            self.hstate[h.name] = hs
        self.fpack = torch.zeros(foff, dtype=torch.bfloat16, device=device)
        self.dpack = torch.zeros(max(doff, 8), dtype=torch.bfloat16, device=device)
        self.pack_descs = torch.tensor(descs, dtype=torch.int64, device=device)
        assert self.lib.kodhip_pack_desc_bytes() == 13 * 8
        self.pack_blocks = blk
        self.exec_units = exec_units
        self.device = device
        self.hyper = torch.zeros(10, dtype=torch.float32, device=device)
        # pinned staging ring: the H2D copy is asynchronous, so a slot is not rewritten for the next 15 uploads
        self._hyper_host = [torch.zeros(10, dtype=torch.float32).pin_memory() for _ in range(16)]
        self._hyper_events = [None] * len(self._hyper_host)
        self._hyper_slot = 0
        self._hyper_vals = None

    def _grad_view(self, name, arena=None):
        o, k = self.layout[name]
        a = self.g_arena[self.g_cur] if arena is None else arena
        return a[o:o + k].view(self.params[name].shape)

    # ------------------------------------------------------------------ activations
    _UNIT_FIELDS = ("stats", "T", "sums", "aff", "bsums", "bsums_g", "bpart", "T2", "coef", "raw", "M", "H", "W", "Ho",
                    "Wo", "fused_red", "segs", "seg_slots", "raw_ld", "wg_splits", "wg_off")
    _HEAD_FIELDS = ("H", "W", "M", "dy", "ws", "wg_splits", "wg_off")

    def _export_set(self) -> dict:
        return dict(act=self.act, gact=self.gact, gact32=self.gact32, wg_part=self.wg_part, pool_idx=self.pool_idx,
                    red_groups=self.red_groups,
                    units={n: {f: getattr(st, f) for f in self._UNIT_FIELDS} for n, st in self.ustate.items()},
                    heads={n: {f: hs[f] for f in self._HEAD_FIELDS} for n, hs in self.hstate.items()})

    def _import_set(self, d: dict):
        self.act, self.gact, self.wg_part, self.pool_idx = d["act"], d["gact"], d["wg_part"], d["pool_idx"]
        self.gact32, self.red_groups = d["gact32"], d["red_groups"]
        for n, fields in d["units"].items():
            st = self.ustate[n]
            for f, v in fields.items():
                setattr(st, f, v)
        for n, fields in d["heads"].items():
            self.hstate[n].update(fields)

    def pin_shape(self, B: int, H: int, W: int):
        """A captured graph replays into the buffer set of this shape: keep it for the engine's lifetime."""
        self._pinned.add((B, H, W))

    def allocate(self, B: int, H: int, W: int):
        """Make the buffer set of (B, H, W) current.  Sets are kept (a dict keyed by shape), never reallocated: a
        forward at another shape swaps pointers and leaves the previous set - and any hipGraph captured over it -
        intact.  Unpinned sets beyond KODHIP_MAX_SHAPE_SETS are dropped least-recently-used first."""
        key = (B, H, W)
        if self.shape == key:
            return
        assert H % 32 == 0 and W % 32 == 0, "image size must be a multiple of 32"
        if self.shape is not None:
            self._sets.pop(self.shape, None)
            self._sets[self.shape] = self._export_set()          # (re-inserted last = most recently used)
        self.training_ready = False                              # a pending backward belongs to the previous set
        if key in self._sets:
            d = self._sets.pop(key)
            self._sets[key] = d
            self._import_set(d)
            self.shape = key
            return
        for old in [k for k in self._sets if k not in self._pinned][:max(0, len(self._sets) + 1 - self.max_shape_sets)]:
            del self._sets[old]
        dev = self.device
        lib = self.lib
        self.shape = key
        self.act: Dict[str, torch.Tensor] = {}
        self.gact: Dict[str, torch.Tensor] = {}
        for b in self.g.bufs:
            h, w = H // b.stride, W // b.stride
            if b.name == "image":
                shp = (B, H, W // 2, 8)
            else:
                shp = (B, h, w, b.C)
            self.act[b.name] = torch.empty(shp, dtype=torch.bfloat16, device=dev)
            if b.name != "image":
                self.gact[b.name] = torch.empty(shp, dtype=torch.bfloat16, device=dev)
        max_part = 0
        for u in self.exec_units:
            st = self.ustate[u.name]
            if u.stem:
                st.H, st.W = H, W // 2
                st.Ho, st.Wo = H // 2, W // 2
            else:
                st.H, st.W = H // u.src.stride, W // u.src.stride
                st.Ho, st.Wo = st.H // u.s, st.W // u.s
            st.M = B * st.Ho * st.Wo
            st.raw = torch.empty((B, st.Ho, st.Wo, u.cout), dtype=torch.bfloat16, device=dev)
            st.raw_ld = u.cout                         # row stride of raw (pre-BN output / dY)
            st.T = lib.kodhip_conv_stats_slots(st.M, u.cout)
            st.stats = torch.empty(2 * u.cout * st.T, dtype=torch.float32, device=dev)
            st.sums = torch.empty(2 * u.cout, dtype=torch.float64, device=dev)
            st.aff = torch.empty(4 * u.cout, dtype=torch.float32, device=dev)        # scale|shift|mean|rstd
            st.T2 = lib.kodhip_bn_bwd_slots(st.M, u.cout)
            st.bpart = torch.empty(2 * u.cout * st.T2, dtype=torch.float32, device=dev)
            st.bsums = torch.empty(2 * u.cout, dtype=torch.float64, device=dev)
            st.bsums_g = torch.empty(2 * u.cout, dtype=torch.float64, device=dev)
            st.coef = torch.empty(3 * u.cout, dtype=torch.float32, device=dev)
            wgeo = (B, st.H, st.W, 8, 8, u.cout, 6, 3, 2, 1, 2, 1) if u.stem else \
                (B, st.H, st.W, u.src.buf.C, u.cin, u.cout, u.k, u.k, u.s, u.s, u.p, u.p)
            st.wg_splits = lib.kodhip_conv_wgrad_splits_geo(*wgeo, st.Kp, u.cout)
            # slab region [splits][cout][Kp] (floats): ONE scratch shared by all layers (reduced right after each weight
            # gradient, while it is still in the 256 MB Infinity Cache) - or, for the per-bucket reduction, a region each
            own = self.opt.wgrad_reduce_batched
            st.wg_off = max_part if own else 0
            max_part = max_part + _pad(st.wg_splits * u.cout * st.Kp) if own else max(max_part, st.wg_splits * u.cout * st.Kp)
        self._plan_bn_fusion(B)
        self.gact32 = {}
        if self._f32plan is not None:
            for name in self._f32plan.shadow_bufs:
                self.gact32[name] = torch.empty(self.gact[name].shape, dtype=torch.float32, device=dev)
        for h in self.g.heads:
            hs = self.hstate[h.name]
            hh, ww = H // h.stride, W // h.stride
            hs.update(H=hh, W=ww, M=B * hh * ww)
            hs["dy"] = torch.empty((B * hh * ww, self.head_npad), dtype=torch.bfloat16, device=dev)
            hs["ws"] = torch.empty(2048 * self.head_npad, dtype=torch.float32, device=dev)
            hs["wg_splits"] = lib.kodhip_conv_wgrad_splits_geo(B, hh, ww, h.src.buf.C, h.cin, self.head_npad, 1, 1, 1, 1, 0, 0,
                                                               hs["Kp"], self.head_npad)
            hs["wg_off"] = max_part if own else 0
            nslab = hs["wg_splits"] * self.head_npad * hs["Kp"]
            max_part = max_part + _pad(nslab) if own else max(max_part, nslab)
        self.wg_part = torch.empty(max_part, dtype=torch.float32, device=dev)
        self._plan_wgrad_reduce()
        # SPPF argmax indices
        self.pool_idx = []
        for op in self.g.ops:
            if op.kind == "pool":
                h, w = H // op.src.stride, W // op.src.stride
                self.pool_idx.append(torch.empty((B, h, w, op.src.C), dtype=torch.uint8, device=dev))

    def _plan_wgrad_reduce(self):
        """Weight-gradient slab reductions, one launch per gradient bucket (csrc/conv_wgrad.hip: wgrad_reduce_batched):
        {trigger unit index: (device descriptor table, n, total blocks)} - the bucket's layers in arena order.  The
        buckets are the all-reduce buckets of the data-parallel path (engine/ddp.py), planned the same way on one GPU."""
        lib = self.lib
        dt = np.dtype([("part_off", "<i8"), ("grad_off", "<i8"), ("splits", "<i4"), ("Nfull", "<i4"), ("N", "<i4"), ("K", "<i4"),
                       ("Kp", "<i4"), ("Cin", "<i4"), ("KK", "<i4"), ("stem", "<i4"), ("scale", "<f4"), ("block_start", "<i4")])
        assert dt.itemsize == lib.kodhip_wgrad_reduce_desc_bytes()
        A, nc = self.g.num_anchors, self.g.num_classes
        layers = []                   # arena order = forward execution order: (weight offset, descriptor fields)
        for u in self.exec_units:
            st = self.ustate[u.name]
            K = 144 if u.stem else u.k * u.k * u.cin
            layers.append((st.w_off, dict(part_off=st.wg_off, grad_off=st.w_off, splits=st.wg_splits, Nfull=u.cout, N=u.cout, K=K,
                                          Kp=st.Kp, Cin=8 if u.stem else u.cin, KK=18 if u.stem else u.k * u.k,
                                          stem=1 if u.stem else 0, scale=1.0)))
        for h in self.g.heads:
            hs = self.hstate[h.name]
            layers.append((hs["w_off"], dict(part_off=hs["wg_off"], grad_off=hs["w_off"], splits=hs["wg_splits"], Nfull=self.head_npad,
                                             N=A * (5 + nc), K=h.cin, Kp=hs["Kp"], Cin=h.cin, KK=1, stem=0, scale=1.0)))
        self.red_groups = {}
        self._red_bucket_bytes = self.bucket_bytes
        for trig, lo, hi in plan_buckets(self.unit_starts, self.n_arena, max(self.bucket_bytes // 4, 1)):
            rows, blk = [], 0
            for off, d in layers:
                if lo <= off < hi:
                    d = dict(d, block_start=blk)
                    blk += lib.kodhip_wgrad_reduce_blocks(d["N"], d["K"])
                    rows.append(tuple(d[k] for k in dt.names))
            if rows:
                arr = np.array(rows, dtype=dt)
                tab = torch.from_numpy(arr.view(np.uint8).reshape(-1).copy()).to(self.device)
                self.red_groups[trig] = (tab, len(rows), blk)

    def _check_equal_local_batch(self, key):
        """SyncBN here divides the all-reduced sums by M_local * world_size (torch's SyncBatchNorm all-gathers the
        per-rank counts instead): that is only right when every rank holds the same number of pixels, so the first
        TRAINING forward of a shape under SyncBN checks it across the group and refuses uneven local batches loudly.
        (Only there: eval forwards exchange nothing, so validation on one rank, or with uneven last batches, must not
        meet a collective.)"""
        if not (self.collectives and self.sync_bn and self.world_size > 1) or key in self._checked_shapes:
            return
        self._checked_shapes.add(key)
        import torch.distributed as dist
        shapes = [None] * self.world_size
        dist.all_gather_object(shapes, tuple(key), group=self.process_group)
        if any(tuple(s) != tuple(key) for s in shapes):
            raise RuntimeError(f"SyncBN needs the same local batch shape on every rank, got {shapes}: pad or drop the "
                               "last uneven batch (DistributedSampler drop_last / padding)")

    def _plan_bn_fusion(self, B: int):
        """Static analysis of the backward program: for every conv unit U find the LAST launch that writes U's
        output-gradient slice before U's own BatchNorm backward.  When that launch is a (FAST-path) data gradient
        whose output covers the slice, it also produces U's BN-backward reduction in its epilogue
        (kodhip_conv_dgrad_bnred) and U skips its own reduce pass."""
        lib = self.lib
        for u in self.exec_units:
            st = self.ustate[u.name]
            st.fused_red, st.segs, st.seg_slots = False, None, 0
        # dual data gradients: a CSP layer's main_conv and short_conv (both pointwise, same input) write dX in ONE launch
        self._dual = {}                # main unit name -> its short_conv unit
        if self.opt.dual_dgrad:
            for u in self.exec_units:
                v = u.sibling
                if (v is not None and u.k == v.k == 1 and u.s == v.s == 1 and u.p == v.p == 0 and u.cout == v.cout
                        and (u.src.buf.name, u.src.coff, u.src.C) == (v.src.buf.name, v.src.coff, v.src.C)
                        and u.cout % 8 == 0):
                    self._dual[u.name] = v
        dual_shorts = {v.name for v in self._dual.values()}
        # activation gradients with several producers: accumulated in fp32 (one rounding) instead of bf16 read-modify-write
        self._f32plan = None
        if self.opt.dx_accum_fp32:
            self._f32plan = plan_f32_accumulation(backward_writes(self.g, dual_shorts)[0], {b.name: b.C for b in self.g.bufs})
            if self.opt.debug_plan:
                print(f"[kodhip] fp32 accumulation of multi-producer gradients: shadows {sorted(self._f32plan.shadow_bufs)}; "
                      f"bf16 (unsupported) {self._f32plan.unsupported}", flush=True)
        if not self.opt.bn_reduce_fused:
            return
        ws, upos = backward_writes(self.g, dual_shorts)          # engine/plan.py: who writes which gradient buffer, in order
        writes = [(w.pos, w.unit, w.buf, w.lo, w.hi) for w in ws]
        plan = {}
        for u in self.exec_units:
            lo, hi = u.dst.coff, u.dst.coff + u.dst.C
            cand = [w for w in writes if w[0] < upos[u.name] and w[2] == u.dst.buf.name and w[3] < hi and w[4] > lo]
            if not cand:
                continue
            last = max(cand, key=lambda w: w[0])
            if last[1] is not None and last[3] <= lo and last[4] >= hi:
                plan.setdefault(last[1], []).append((u, lo - last[3]))
        units = {u.name: u for u in self.exec_units}
        for wname, prods in plan.items():
            w = units[wname]
            ws = self.ustate[wname]
            s2 = int(w.k == 3 and w.s == 2 and w.p == 1)
            # (A/B knob: only fuse into launches whose reduction length is at least KODHIP_BNRED_MINK; measured best: all)
            if w.k * w.k * w.cout < self.opt.bn_reduce_min_k:
                continue
            if wname in self._dual:
                slots = lib.kodhip_conv_dgrad_dual_bnred_slots(B, ws.H, ws.W, w.cin, w.cout, w.cout)
            else:
                if s2 and ws.s2_fold:
                    slots = lib.kodhip_conv_dgrad_s2f_bnred_slots(B, ws.H, ws.W, w.cin, w.cout, w.cout)
                else:
                    slots = lib.kodhip_conv_dgrad_bnred_slots(B, ws.H, ws.W, w.cin, w.cout, w.k, w.k, w.s, w.s, w.p, w.p, w.cout, s2)
            if slots <= 0:
                continue
            prods = prods[:3]                        # MAX_SEG of the kernel
            segs = (_lib.KodBnRedSeg * len(prods))()
            for i, (u, ch0) in enumerate(prods):
                st = self.ustate[u.name]
                st.fused_red, st.T2 = True, slots
                st.bpart = torch.empty(2 * u.cout * slots, dtype=torch.float32, device=self.device)
                segs[i].ch_begin, segs[i].ch_count = ch0, u.cout
                segs[i].raw, segs[i].ldr = st.raw.data_ptr(), u.cout
                segs[i].aff, segs[i].partials = st.aff.data_ptr(), st.bpart.data_ptr()
            ws.segs, ws.seg_slots = segs, slots
        if self.opt.debug_plan:
            fused = [u.name for u in self.exec_units if self.ustate[u.name].fused_red]
            print(f"[kodhip] BN-backward reduction fused into a data gradient for {len(fused)} of {len(self.exec_units)} units; "
                  f"separate pass: {[u.name for u in self.exec_units if not self.ustate[u.name].fused_red]}", flush=True)

    # ------------------------------------------------------------------ helpers
    def _ptr(self, v: View, grad=False):
        t = (self.gact if grad else self.act)[v.buf.name]
        return t.data_ptr()

    def _stream(self):
        return torch.cuda.current_stream().cuda_stream

    # -- per-family kernel timing (bench.py's roofline table): HIP events around every launch of an eager step, recorded
    #    on the stream the launch goes to.  self.profile = [] switches it on; entries (family, e0, e1, algorithmic bytes).
    def _t0(self, stream=None):
        if self.profile is None:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream) if stream is not None else e.record()
        return e

    def _t1(self, e0, family: str, nbytes: float, stream=None):
        if e0 is None:
            return
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record(stream) if stream is not None else e1.record()
        self.profile.append((family, e0, e1, nbytes))

    def pack_weights(self):
        _lib.check(self.lib.kodhip_pack_weights(self.p_arena.data_ptr(), self.fpack.data_ptr(),
                                                self.dpack.data_ptr(), self.pack_descs.data_ptr(),
                                                self.pack_descs.shape[0], self.pack_blocks, self._stream()),
                   "pack_weights")
        self._packed_version = self.param_version

    def _allreduce(self, t):
        if self.collectives:
            if self.comm is not None:
                self.comm.all_reduce(t)
            else:
                torch.distributed.all_reduce(t, group=self.process_group)

    def _allreduce_group(self, tensors, outs=None):
        """In-place (or, with `outs`, out-of-place) sum all-reduce of several small tensors as ONE collective launch
        (ncclGroupStart / End) on the native communicator; one call each on a torch.distributed group."""
        if not self.collectives:
            return
        if self.comm is not None:
            if len(tensors) == 1:
                self.comm.all_reduce(tensors[0]) if outs is None else self.comm.all_reduce_to(tensors[0], outs[0])
                return
            with self.comm.group():
                for k, t in enumerate(tensors):
                    self.comm.all_reduce(t) if outs is None else self.comm.all_reduce_to(t, outs[k])
        else:
            for k, t in enumerate(tensors):
                if outs is not None:
                    outs[k].copy_(t)
                    t = outs[k]
                torch.distributed.all_reduce(t, group=self.process_group)

    # ------------------------------------------------------------------ forward
    def forward(self, x: torch.Tensor, training: bool = True, after_first_layer=None):
        """x: [B,3,H,W] fp32 NCHW on this device.  Returns 3 tensors [B,A,h,w,5+nc] fp32 (ll, ml, hl).
        after_first_layer: called once after the first layer's kernels are launched (a hook for side-stream work that only
        depends on the step's inputs: it is then captured behind the forward chain's head, see Yolov5Network.train_step)."""
        lib, chk = self.lib, _lib.check
        B, Cimg, H, W = x.shape
        assert Cimg == 3 and x.dtype == torch.float32 and x.is_contiguous() and x.device == self.device
        self.allocate(B, H, W)
        s = self._stream()
        if self._packed_version != self.param_version:
            self.pack_weights()
        chk(lib.kodhip_nchw_to_nhwc4(x.data_ptr(), self.act["image"].data_ptr(), B, 3, H, W, s), "nchw_to_nhwc4")
        A, nc = self.g.num_anchors, self.g.num_classes
        outs = []
        pool_i = 0
        fp, pa = self.fpack.data_ptr(), self.p_arena.data_ptr()
        eval_aff = None if training else self._eval_affine_ptrs()
        sync = training and self.sync_bn and self.collectives
        if sync:
            self._check_equal_local_batch((B, H, W))
        rm, rv = self.rm_arena.data_ptr(), self.rv_arena.data_ptr()

        def conv_stage(u: ConvUnit, s=s):
            st, C_ = self.ustate[u.name], u.cout
            if u.stem:
                geo = (B, st.H, st.W, 8, 0, 32, C_, 6, 1, 2, 1, 2, 1, st.Kp_f)      # wide-pixel form, see Kp_f
            else:
                geo = (B, st.H, st.W, u.src.buf.C, u.src.coff, u.cin, C_, u.k, u.k, u.s, u.s, u.p, u.p, st.Kp_f)
            e0 = self._t0()
            chk(lib.kodhip_conv_fwd_raw(self._ptr(u.src), fp + 2 * st.f_off, st.raw.data_ptr(),
                                        st.stats.data_ptr(), *geo, C_, 0, s), u.name)
            cin_true = 3 if u.stem else u.cin
            in_px = B * H * W if u.stem else B * st.H * st.W
            self._t1(e0, "conv_fwd", 2 * (in_px * cin_true + st.M * C_))

        def stats_stage(group, s=s):
            """Batch statistics -> BatchNorm constants.  Under SyncBN the [sum, sum of squares] vectors of the group's
            units (a CSP layer's main + short convs) are exchanged as ONE grouped collective."""
            e0 = self._t0()
            if not sync:
                for u in group:
                    st, C_ = self.ustate[u.name], u.cout
                    aff = st.aff.data_ptr()
                    chk(lib.kodhip_bn_finalize_partials(st.stats.data_ptr(), st.T, float(st.M), pa + 4 * st.g_off,
                                                        pa + 4 * st.b_off, rm + 4 * st.rs_off, rv + 4 * st.rs_off,
                                                        BN_MOMENTUM, BN_EPS, aff, aff + 4 * C_, aff + 8 * C_,
                                                        aff + 12 * C_, C_, 1, s), u.name)
            elif self.peer is not None:
                # SyncBN over peer buffers: the same single launch per unit, the ranks' sums meet inside the kernel
                for u in group:
                    st, C_ = self.ustate[u.name], u.cout
                    aff = st.aff.data_ptr()
                    chk(lib.kodhip_bn_finalize_partials_peer(st.stats.data_ptr(), st.T, float(st.M) * self.world_size,
                                                             pa + 4 * st.g_off, pa + 4 * st.b_off, rm + 4 * st.rs_off,
                                                             rv + 4 * st.rs_off, BN_MOMENTUM, BN_EPS, aff, aff + 4 * C_,
                                                             aff + 8 * C_, aff + 12 * C_, C_, 1, self.peer.view_ptr(),
                                                             self.peer_slots[(u.name, "f")], s), u.name)
            else:
                for u in group:
                    st = self.ustate[u.name]
                    chk(lib.kodhip_bn_reduce_partials(st.stats.data_ptr(), st.sums.data_ptr(), u.cout, st.T, s), u.name)
                self._allreduce_group([self.ustate[u.name].sums for u in group])
                for u in group:
                    st, C_ = self.ustate[u.name], u.cout
                    aff = st.aff.data_ptr()
                    chk(lib.kodhip_bn_finalize(st.sums.data_ptr(), float(st.M) * self.world_size, pa + 4 * st.g_off,
                                               pa + 4 * st.b_off, rm + 4 * st.rs_off, rv + 4 * st.rs_off, BN_MOMENTUM,
                                               BN_EPS, aff, aff + 4 * C_, aff + 8 * C_, aff + 12 * C_, C_, 1, s), u.name)
            self._t1(e0, "bn_finalize", sum(8.0 * u.cout * self.ustate[u.name].T for u in group))

        def apply_stage(u: ConvUnit, s=s):
            st, C_ = self.ustate[u.name], u.cout
            aff = st.aff.data_ptr()
            sc_p, sh_p = (aff, aff + 4 * C_) if training else eval_aff[u.name]
            res = u.residual
            e0 = self._t0()
            chk(lib.kodhip_bn_silu_apply(st.raw.data_ptr(), st.raw_ld, sc_p, sh_p,
                                         self._ptr(res) if res else None, res.buf.C if res else 0,
                                         res.coff if res else 0,
                                         self._ptr(u.dst), u.dst.buf.C, u.dst.coff, st.M, C_, s), u.name)
            self._t1(e0, "bn_silu_apply", (6.0 if res else 4.0) * st.M * C_)

        # A CSP layer's short_conv (conv -> statistics -> apply) depends only on the layer input and is needed only by
        # last_conv: it runs on a side stream next to main_conv and the blocks, where it fills the chip while the main
        # branch sits in a single-block statistics kernel or a latency-bound deep layer.  (Not under SyncBN - the two
        # statistic exchanges travel as one grouped collective on the main stream - and not while timing families.)
        main_stream = torch.cuda.current_stream()
        if sync and self.peer is not None:
            self.peer.step_begin(s)            # the step's sequence number: tags every statistic this rank publishes
        # (with the peer exchange there is no communicator whose call order the side streams could disturb)
        branch = training and (not sync or self.peer is not None) and self.branch_overlap and self.profile is None
        # the P3 / P4 head convolutions are leaves (only the loss reads them): they run on their own side stream as soon
        # as their input exists, beside the bottom-up path, instead of after it.  head_src: buffer -> "ready" event
        heads_aside = training and self.branch_overlap and self.profile is None          # (also under SyncBN: no collective involved)
        head_src = {op.src.buf.name: None for op in self.g.ops[:-1] if op.kind == "head"} if heads_aside else {}
        heads_on_aux = False
        joined_buf = None                # concat buffer whose short_conv half is being written on the side stream
        ops = self.g.ops
        i = 0
        while i < len(ops):
            op = ops[i]
            i += 1
            if op.kind == "conv" and joined_buf is not None and op.unit.src.buf.name == joined_buf:
                main_stream.wait_stream(self.br_stream)
                joined_buf = None
            if op.kind == "conv" and branch and op.unit.sibling is not None and i < len(ops) and \
                    ops[i].unit is op.unit.sibling and joined_buf is None:
                short = ops[i].unit
                i += 1
                if self.br_stream is None:
                    self.br_stream = torch.cuda.Stream(device=self.device)
                # the fork's dependency is taken here, the side branch is CAPTURED after the main branch's kernels: the
                # graph executor keeps a node's first captured successor on its queue (see backward())
                fork = torch.cuda.Event()
                fork.record(main_stream)
                conv_stage(op.unit)
                stats_stage([op.unit])
                apply_stage(op.unit)
                self.br_stream.wait_event(fork)
                bs = self.br_stream.cuda_stream
                conv_stage(short, bs)
                stats_stage([short], bs)
                apply_stage(short, bs)
                joined_buf = short.dst.buf.name
                continue
            if op.kind == "conv" and op.unit.dst.buf.name in head_src and not (branch and op.unit.sibling is not None):
                conv_stage(op.unit)
                stats_stage([op.unit])
                apply_stage(op.unit)
                ev = torch.cuda.Event()
                ev.record(main_stream)
                head_src[op.unit.dst.buf.name] = ev
                continue
            if op.kind == "conv" and after_first_layer is not None and i > 1:
                after_first_layer()
                after_first_layer = None
            if op.kind == "conv":
                group = [op.unit]
                # SyncBN over RCCL: a unit and its sibling (same input, next in the program) share one statistic exchange
                if sync and self.peer is None and op.unit.sibling is not None and i < len(ops) and ops[i].unit is op.unit.sibling:
                    group.append(ops[i].unit)
                    i += 1
                for u in group:
                    conv_stage(u)
                if training:
                    stats_stage(group)
                for u in group:
                    apply_stage(u)
            elif op.kind == "pool":
                h, w = H // op.src.stride, W // op.src.stride
                chk(lib.kodhip_maxpool5_fwd(self._ptr(op.src), op.src.buf.C, op.src.coff, self._ptr(op.dst),
                                            op.dst.buf.C, op.dst.coff, self.pool_idx[pool_i].data_ptr(),
                                            B, h, w, op.src.C, s), "maxpool")
                pool_i += 1
            elif op.kind == "up":
                h, w = H // op.src.stride, W // op.src.stride
                chk(lib.kodhip_upsample2x_fwd(self._ptr(op.src), op.src.buf.C, op.src.coff, self._ptr(op.dst),
                                              op.dst.buf.C, op.dst.coff, B, h, w, op.src.C, s), "upsample")
            else:
                hu: HeadUnit = op.unit
                hs = self.hstate[hu.name]
                out = torch.empty((B, A, hs["H"], hs["W"], 5 + nc), dtype=torch.float32, device=self.device)
                hstream = s
                ev = head_src.get(hu.src.buf.name)
                if ev is not None:
                    if self.head_stream is None:
                        self.head_stream = torch.cuda.Stream(device=self.device)
                    self.head_stream.wait_event(ev)
                    hstream, heads_on_aux = self.head_stream.cuda_stream, True
                chk(lib.kodhip_conv_fwd_head(self._ptr(hu.src), fp + 2 * hs["f_off"], pa + 4 * hs["b_off"],
                                             out.data_ptr(), B, hs["H"], hs["W"], hu.src.buf.C, hu.src.coff,
                                             hu.cin, A, nc, hs["Kp"], hstream), hu.name)
                outs.append(out)
        if after_first_layer is not None:
            after_first_layer()
        if joined_buf is not None:
            main_stream.wait_stream(self.br_stream)
        if heads_on_aux:
            main_stream.wait_stream(self.head_stream)
        if training:
            self.nbt_arena += 1
            self.stats_version += 1              # running statistics moved
        self.training_ready = training          # an eval forward overwrites the saved pre-BN tensors
        return outs

    def _eval_affine_ptrs(self):
        """Eval-mode BatchNorm constants of every unit (scale = gamma * rsqrt(running_var + eps), shift = beta -
        running_mean * scale) in ONE flat buffer, recomputed with five whole-network tensor ops only when parameters or
        running statistics changed - not per layer per forward (a validation epoch forwards many batches with frozen
        weights).  Kept apart from the training constants (st.aff), so an eval forward never disturbs a pending backward.
        Returns {unit name: (scale ptr, shift ptr)}."""
        # keyed on the arenas' own version counters too: in-place edits that bypass the engine (EMA swap,
        # reset_running_stats, a non-fused optimizer, a user-captured graph replay bumps nothing - see invalidate_eval_constants)
        key = (self.param_version, self.stats_version, self.p_arena._version, self.rm_arena._version, self.rv_arena._version)
        if self._eval_aff is None:
            gi, bi, ri, off = [], [], [], 0
            self._eval_off = {}
            for u in self.exec_units:
                st = self.ustate[u.name]
                ar = torch.arange(u.cout)
                gi.append(st.g_off + ar); bi.append(st.b_off + ar); ri.append(st.rs_off + ar)
                self._eval_off[u.name] = off
                off += u.cout
            dev = self.device
            self._eval_idx = tuple(torch.cat(t).to(dev) for t in (gi, bi, ri))
            self._eval_n = off
            self._eval_aff = torch.empty(2 * off, dtype=torch.float32, device=dev)
            self._eval_key = None
        if self._eval_key != key:
            gi, bi, ri = self._eval_idx
            n = self._eval_n
            sc = self.p_arena[gi] * torch.rsqrt(self.rv_arena[ri] + BN_EPS)
            self._eval_aff[:n] = sc
            self._eval_aff[n:] = self.p_arena[bi] - self.rm_arena[ri] * sc
            self._eval_key = key
        base, n = self._eval_aff.data_ptr(), self._eval_n
        return {name: (base + 4 * o, base + 4 * (n + o)) for name, o in self._eval_off.items()}

    # ------------------------------------------------------------------ backward
    def backward(self, head_grads: List[torch.Tensor]):
        """head_grads: d loss / d (ll, ml, hl) head tensors.  Fills the gradient arena; returns nothing."""
        assert self.training_ready, "backward() needs a preceding training forward()"
        self.training_ready = False
        lib, chk = self.lib, _lib.check
        B, H, W = self.shape
        s = self._stream()
        A, nc = self.g.num_anchors, self.g.num_classes
        ga = self.g_arena[self.g_cur]
        gp = ga.data_ptr()
        fp, dp = self.fpack.data_ptr(), self.dpack.data_ptr()
        pa = self.p_arena.data_ptr()
        wgp = self.wg_part.data_ptr()
        touched = set()            # grad buffers already holding a (partial) sum
        # Weight gradients run on a side stream: dW of a layer is off the critical path (bn-bwd -> dgrad -> next
        # layer), so it fills the tails of the small kernels on the main stream and, under SyncBN, the latency of
        # the per-layer statistic all-reduce.  All wgrads share one stream (and the split-K scratch) => ordered.
        main = torch.cuda.current_stream()
        wg = None
        if self.wgrad_overlap:
            if self.wg_stream is None:
                self.wg_stream = torch.cuda.Stream(device=self.device)
            wg = self.wg_stream

        # How a weight gradient joins the side stream matters in the captured graph: this stack's graph executor keeps a
        # node's FIRST captured successor on the node's queue and hands the later ones to other queues (~11 us per
        # hand-over).  So a weight gradient takes its dependency where dY is ready (an event right after
        # bn_silu_bwd_apply / head_bwd_prep - it then runs beside the same unit's data gradient, both reading dY) but
        # is launched, i.e. captured, only after the main stream's next kernel (the data gradient): the critical chain
        # apply -> dgrad -> next unit's coefficients -> ... stays on one queue and only the off-path weight gradients
        # pay the hand-over.  KODHIP_WGRAD_FORK=legacy: wait_stream at the call site, behind the data gradient (round 1).
        deferred = []                  # [(event, name, nbytes, args)]
        defer = wg is not None and self.wgrad_fork != "legacy"

        def fork_point(stream=None):
            """call right after the kernel that completes dY (on `stream`, default the main stream)"""
            if defer:
                self._fork_ev = torch.cuda.Event()
                self._fork_ev.record(stream or main)
        self._fork_point = fork_point

        batched = self.opt.wgrad_reduce_batched     # slab reductions: one launch per bucket (default) | per layer

        def launch_wgrad(name, nbytes, args, stream_obj):
            """args = kodhip_conv_wgrad's (x, dy, slab region, grad, geometry ..., n_valid, stem, scale)"""
            e0 = self._t0(stream_obj)
            sid = stream_obj.cuda_stream if stream_obj is not None else s
            if batched:
                chk(lib.kodhip_conv_wgrad_partial(*args[:3], *args[4:-3], sid), name + ".wgrad")
            else:
                chk(lib.kodhip_conv_wgrad(*args, sid), name + ".wgrad")
            self._t1(e0, "wgrad", nbytes, stream_obj)

        def flush_wgrads():
            """call after the main stream's next kernel has been launched"""
            for ev, name, nbytes, args in deferred:
                wg.wait_event(ev)
                launch_wgrad(name, nbytes, args, wg)
            deferred.clear()
            if due:
                self._launch_due()
        self._flush_wgrads = flush_wgrads

        def timed_wgrad(name, nbytes, *args):
            if defer:
                ev, self._fork_ev = self._fork_ev, None
                if ev is None:
                    ev = torch.cuda.Event()
                    ev.record(main)
                deferred.append((ev, name, nbytes, args))
                return
            if wg is not None:
                wg.wait_stream(main)
            launch_wgrad(name, nbytes, args, wg)

        # gradient buffers last written on a side stream (the P3 / P4 heads' data gradients): buffer -> event the main
        # stream must wait for before it reads or accumulates into the buffer
        grad_events = {}

        def sync_grad(name):
            ev = grad_events.pop(name, None)
            if ev is not None:
                main.wait_event(ev)
        self._sync_grad = sync_grad

        def acc_flag(v: View) -> int:
            """0 = first writer (overwrite), 1 = accumulate; zero-fills on a partial first touch."""
            name = v.buf.name
            sync_grad(name)
            if name in touched:
                return 1
            touched.add(name)
            if v.C != v.buf.C:
                self.gact[name].zero_()
                if name in self.gact32:
                    self.gact32[name].zero_()
                return 1
            return 0

        op_index = {id(o): i for i, o in enumerate(self.g.ops)}

        def f32(kind, ident, v: View):
            """(bits 8.. of the `accumulate` argument, fp32 shadow pointer) of one gradient-buffer write (engine/plan.py)"""
            if self._f32plan is None:
                return 0, None
            mode = self._f32plan.modes.get((kind, op_index[id(ident)] if kind in ("up", "pool") else ident), 0)
            sh = self.gact32.get(v.buf.name)
            return mode << 8, (sh.data_ptr() if (sh is not None and mode in (1, 2, 3)) else None)
        self._f32 = f32

        self._pending = []
        if self._red_bucket_bytes != self.bucket_bytes:      # the reductions follow the all-reduce buckets
            self._plan_wgrad_reduce()
        buckets = {}
        if self.collectives:
            buckets = {trig: (lo, hi) for trig, lo, hi in plan_buckets(self.unit_starts, self.n_arena,
                                                                        max(self.bucket_bytes // 4, 1))}
        unit_i = len(self.unit_starts)
        pool_i = len(self.pool_idx)
        head_i = len(self.g.heads)
        sync = self.sync_bn and self.collectives
        rccl_sync = sync and self.peer is None

        due = []                       # gradient buckets whose last unit has been processed: launched at the next flush point

        def launch_due():
            for idx in due:
                if batched and idx in self.red_groups:       # (KODHIP_WGRAD_REDUCE=bucket) reduce all the bucket's slabs at once
                    tab, n_desc, blocks = self.red_groups[idx]
                    e0 = self._t0(wg)
                    chk(lib.kodhip_wgrad_reduce_batched(wgp, gp, tab.data_ptr(), n_desc, blocks,
                                                        wg.cuda_stream if wg is not None else s), "wgrad_reduce_batched")
                    self._t1(e0, "wgrad", 0.0, wg)
                if idx in buckets:
                    lo, hi = buckets[idx]
                    cs = self._comm_stream()
                    # overlapped buckets use their own communicator: SyncBN sums (main stream) and buckets (side stream)
                    # never interleave on one communicator from two streams
                    bc = self.comm_buckets if (cs is not None and self.comm_buckets is not None) else self.comm
                    # on the weight-gradient stream the bucket's last weight gradient has already waited for an event
                    # recorded behind every BatchNorm / bias gradient of the bucket (fork_point): no new edge from the main chain
                    self._pending.append(launch_bucket(ga, lo, hi, self.process_group, cs, bc, also_after=wg,
                                                       wait_caller=not (cs is not None and cs is wg and defer)))
            due.clear()
        self._launch_due = launch_due

        def bucket_tick():
            """one conv / head unit's gradients are complete: buckets finish from the arena's end toward its start.  The
            bucket is launched at a flush point of the weight-gradient stream, never ahead of one: a fused short_conv's
            weight gradient is still deferred here (it is captured behind its main_conv's data gradient, so that the main
            chain's next kernel stays the first captured successor - see flush_wgrads), and flushing it early for the
            bucket's sake moves the main chain to another queue in the replayed graph (measured: -11 % step rate)."""
            nonlocal unit_i
            unit_i -= 1
            if (batched and unit_i in self.red_groups) or unit_i in buckets:
                due.append(unit_i)
                if not deferred:
                    launch_due()

        def bn_bwd_stats(group):
            """BatchNorm-backward sums -> coefficients.  Under SyncBN the [sum dz, sum dz*xhat] vectors of the group's
            units (a CSP layer's short + main convs) are exchanged as ONE grouped collective."""
            for u in group:
                sync_grad(u.dst.buf.name)          # (a head's data gradient on the side stream may be its last writer)
            for u in group:
                st, C_ = self.ustate[u.name], u.cout
                if not st.fused_red:
                    aff, dA = st.aff.data_ptr(), u.dst
                    e0 = self._t0()
                    chk(lib.kodhip_bn_silu_bwd_reduce(self._ptr(dA, True), dA.buf.C, dA.coff, st.raw.data_ptr(), st.raw_ld,
                                                      aff, aff + 4 * C_, aff + 8 * C_, aff + 12 * C_,
                                                      st.bpart.data_ptr(), st.M, C_, s), u.name)
                    self._t1(e0, "bn_bwd_reduce", 4.0 * st.M * C_)
            e0 = self._t0()
            if sync and self.peer is not None:
                for u in group:
                    st, C_ = self.ustate[u.name], u.cout
                    aff = st.aff.data_ptr()
                    chk(lib.kodhip_bn_bwd_coeffs_partials_peer(st.bpart.data_ptr(), st.T2, float(st.M) * self.world_size,
                                                               pa + 4 * st.g_off, aff + 8 * C_, aff + 12 * C_,
                                                               gp + 4 * st.g_off, gp + 4 * st.b_off, st.coef.data_ptr(), C_,
                                                               1 if st.fused_red else 0, self.peer.view_ptr(),
                                                               self.peer_slots[(u.name, "b")], s), u.name)
                self._t1(e0, "bn_bwd_coeffs", sum(8.0 * u.cout * self.ustate[u.name].T2 for u in group))
                return
            if sync:
                for u in group:
                    st = self.ustate[u.name]
                    chk(lib.kodhip_bn_reduce_partials(st.bpart.data_ptr(), st.bsums.data_ptr(), u.cout, st.T2, s), u.name)
                # out of place: the local sums stay for dgamma / dbeta
                self._allreduce_group([self.ustate[u.name].bsums for u in group], [self.ustate[u.name].bsums_g for u in group])
            if not sync and len(group) == 2:           # short_conv + main_conv: one launch for both coefficient sets
                args = []
                for u in group:
                    st, C_ = self.ustate[u.name], u.cout
                    aff = st.aff.data_ptr()
                    args += [st.bpart.data_ptr(), st.T2, float(st.M), pa + 4 * st.g_off, aff + 8 * C_, aff + 12 * C_,
                             gp + 4 * st.g_off, gp + 4 * st.b_off, st.coef.data_ptr(), C_, 1 if st.fused_red else 0]
                chk(lib.kodhip_bn_bwd_coeffs_partials2(*args, s), group[0].name + "+" + group[1].name)
                group_done = True
            else:
                group_done = False
            for u in ([] if group_done else group):
                st, C_ = self.ustate[u.name], u.cout
                aff = st.aff.data_ptr()
                rawm = 1 if st.fused_red else 0        # partials came from the last dgrad into this tensor
                if sync:
                    chk(lib.kodhip_bn_bwd_coeffs(st.bsums.data_ptr(), st.bsums_g.data_ptr(),
                                                 float(st.M) * self.world_size, pa + 4 * st.g_off,
                                                 aff + 8 * C_, aff + 12 * C_, gp + 4 * st.g_off, gp + 4 * st.b_off,
                                                 st.coef.data_ptr(), C_, rawm, s), u.name)
                else:
                    chk(lib.kodhip_bn_bwd_coeffs_partials(st.bpart.data_ptr(), st.T2, float(st.M), pa + 4 * st.g_off,
                                                          aff + 8 * C_, aff + 12 * C_, gp + 4 * st.g_off,
                                                          gp + 4 * st.b_off, st.coef.data_ptr(), C_, rawm, s), u.name)
            self._t1(e0, "bn_bwd_coeffs", sum(8.0 * u.cout * self.ustate[u.name].T2 for u in group))

        # (with every collective on the main stream - KODHIP_COMM_OVERLAP=0, RCCL SyncBN - the head chains stay there too)
        heads_side = (wg is not None and defer and self.branch_overlap and self.profile is None and
                      (not self.collectives or (self._comm_stream() is not None and not rccl_sync)))
        bwd_start = torch.cuda.Event()
        if heads_side:
            bwd_start.record(main)
        rops = list(reversed(self.g.ops))
        ri = 0
        while ri < len(rops):
            op = rops[ri]
            ri += 1
            if op.kind == "head":
                head_i -= 1
                hu: HeadUnit = op.unit
                hs = self.hstate[hu.name]
                gten = head_grads[head_i].contiguous()
                assert gten.shape == (B, A, hs["H"], hs["W"], 5 + nc) and gten.dtype == torch.float32
                names = [f"{hu.name}.{k}_head.conv.bias" for k in ("box", "obj", "cls")]
                offs = [self.layout[n][0] for n in names]
                src = hu.src
                # The three head chains (gradient re-layout -> data gradient) are independent until the neck: the P5
                # chain, which the first backward layers wait for, stays on the main stream; the P4 and P3 chains
                # run beside it on a side stream and the main stream joins each where that level's gradient buffer
                # is next touched (acc_flag / the producing unit's apply).
                side = (heads_side and head_i < len(self.g.heads) - 1 and src.C == src.buf.C and src.buf.name not in touched)
                hstream, hs_ = main, s
                if side:
                    if self.head_stream is None:
                        self.head_stream = torch.cuda.Stream(device=self.device)
                    hstream, hs_ = self.head_stream, self.head_stream.cuda_stream
                    hstream.wait_event(bwd_start)
                chk(lib.kodhip_head_bwd_prep(gten.data_ptr(), hs["dy"].data_ptr(), hs["ws"].data_ptr(),
                                             gp + 4 * offs[0], gp + 4 * offs[1], gp + 4 * offs[2],
                                             B, hs["H"] * hs["W"], A, nc, self.head_npad, hs_), hu.name)
                fork_point(hstream)
                acc = acc_flag(src)
                fm, fptr = f32("head", hu.name, src)
                e0 = self._t0()
                chk(lib.kodhip_conv_dgrad(hs["dy"].data_ptr(), dp + 2 * hs["d_off"], self._ptr(src, True),
                                          B, hs["H"], hs["W"], src.buf.C, src.coff, hu.cin,
                                          self.head_npad, 1, 1, 1, 1, 0, 0, hs["Kdp"], self.head_npad, 0,
                                          acc | fm, fptr, hs_), hu.name + ".dgrad")
                self._t1(e0, "dgrad", 2.0 * hs["M"] * (self.head_npad + hu.cin))
                if side:
                    ev = torch.cuda.Event()
                    ev.record(hstream)
                    grad_events[src.buf.name] = ev
                timed_wgrad(hu.name, 2.0 * hs["M"] * (hu.cin + self.head_npad),
                            self._ptr(src), hs["dy"].data_ptr(), wgp + 4 * hs["wg_off"], gp + 4 * hs["w_off"],
                            B, hs["H"], hs["W"], src.buf.C, src.coff, hu.cin,
                            self.head_npad, 1, 1, 1, 1, 0, 0, hs["Kp"], self.head_npad, 0, A * (5 + nc), 0, 1.0)
                flush_wgrads()
            elif op.kind == "up":
                h, w = H // op.src.stride, W // op.src.stride
                chk(lib.kodhip_upsample2x_bwd(self._ptr(op.dst, True), op.dst.buf.C, op.dst.coff,
                                              self._ptr(op.src, True), op.src.buf.C, op.src.coff,
                                              acc_flag(op.src), B, h, w, op.src.C, f32("up", op, op.src)[1], s), "upsample_bwd")
            elif op.kind == "pool":
                pool_i -= 1
                h, w = H // op.src.stride, W // op.src.stride
                # src and dst are slices of the same (already initialised) concat gradient buffer
                chk(lib.kodhip_maxpool5_bwd(self._ptr(op.dst, True), op.dst.buf.C, op.dst.coff,
                                            self.pool_idx[pool_i].data_ptr(), self._ptr(op.src, True),
                                            op.src.buf.C, op.src.coff, B, h, w, op.src.C, f32("pool", op, op.src)[1], s), "maxpool_bwd")
            else:
                group = [op.unit]
                # SyncBN: short_conv (reached first in reverse order) and its main_conv share one exchange - main's
                # output gradient is complete by now (everything between them in the forward program ran backward)
                if ri < len(rops) and rops[ri].kind == "conv" and rops[ri].unit.sibling is op.unit and \
                        (rccl_sync or rops[ri].unit.name in self._dual):
                    group.append(rops[ri].unit)
                    ri += 1
                bn_bwd_stats(group)
                dual = len(group) == 2 and group[1].name in self._dual          # [short, main]: one data-gradient launch
                for u in group:
                    self._bwd_unit(u, B, H, W, s, gp, pa, dp, wgp, acc_flag, timed_wgrad,
                                   dgrad="skip" if (dual and u is group[0]) else ("dual" if dual else "own"),
                                   partner=group[0] if dual else None)
                    bucket_tick()
                continue
            # gradient buckets complete from the arena's end toward its start
            if op.kind == "head":
                bucket_tick()
        flush_wgrads()
        for name in list(grad_events):
            sync_grad(name)
        if wg is not None:
            main.wait_stream(wg)
        self._publish_grads()

    def _bwd_unit(self, u, B, H, W, s, gp, pa, dp, wgp, acc_flag, timed_wgrad, dgrad="own", partner=None):
        """bn/silu backward apply -> data gradient -> weight gradient of one conv unit (coefficients already in st.coef).
        dgrad: "own" = this unit's launch; "skip" = none (a fused short_conv: its main_conv's launch covers it);
        "dual" = one launch for this unit and `partner` (kodhip_conv_dgrad_dual)."""
        lib, chk = self.lib, _lib.check
        st = self.ustate[u.name]
        C_ = u.cout
        aff = st.aff.data_ptr()
        dA = u.dst
        res = u.residual
        racc = acc_flag(res) if res else 0
        e0 = self._t0()
        chk(lib.kodhip_bn_silu_bwd_apply(self._ptr(dA, True), dA.buf.C, dA.coff, st.raw.data_ptr(), st.raw_ld,
                                         aff, aff + 4 * C_, st.coef.data_ptr(),
                                         self._ptr(res, True) if res else None,
                                         res.buf.C if res else 0, res.coff if res else 0,
                                         racc, st.M, C_, s), u.name)
        self._t1(e0, "bn_silu_bwd_apply", (6.0 + ((4.0 if racc else 2.0) if res else 0.0)) * st.M * C_)
        self._fork_point()
        # st.raw now holds dY
        if u.stem:
            geo = (B, st.H, st.W, 8, 0, 8, C_, 6, 3, 2, 1, 2, 1)
        else:
            geo = (B, st.H, st.W, u.src.buf.C, u.src.coff, u.cin, C_, u.k, u.k, u.s, u.s, u.p, u.p)
            fz = () if st.segs is None else (C.cast(st.segs, C.c_void_p), len(st.segs), st.seg_slots)
            if dgrad == "skip":
                timed_wgrad(u.name, 2.0 * (B * st.H * st.W * u.cin + st.M * C_),
                            self._ptr(u.src), st.raw.data_ptr(), wgp + 4 * st.wg_off, gp + 4 * st.w_off,
                            *geo, st.Kp, C_, 0, C_, 0, 1.0)
                return
            fm, fptr = self._f32("dgrad", u.name, u.src)
            acc_src = acc_flag(u.src) | fm
            in_px = B * st.H * st.W
            # dY read once, dX written once (+ read when accumulating), + the re-read of the producers' pre-BN
            # tensors when this launch carries their BatchNorm-backward reduction
            nb = 2.0 * st.M * C_ + (4.0 if acc_src & 1 else 2.0) * in_px * u.cin
            if st.segs is not None:
                nb += 2.0 * in_px * sum(sg.ch_count for sg in st.segs)
            e0 = self._t0()
            if dgrad == "dual":
                ps = self.ustate[partner.name]
                nb += 2.0 * ps.M * partner.cout
                fn = lib.kodhip_conv_dgrad_dual if st.segs is None else lib.kodhip_conv_dgrad_dual_bnred
                chk(fn(st.raw.data_ptr(), dp + 2 * st.d_off, ps.raw.data_ptr(), dp + 2 * ps.d_off, self._ptr(u.src, True),
                       B, st.H, st.W, u.src.buf.C, u.src.coff, u.cin, C_, st.Kdp, C_, 0, acc_src, fptr, *fz, s), u.name + ".dgrad2")
            elif u.k == 3 and u.s == 2 and u.p == 1:
                if st.s2_fold:
                    fn = lib.kodhip_conv_dgrad_s2f if st.segs is None else lib.kodhip_conv_dgrad_s2f_bnred
                else:
                    fn = lib.kodhip_conv_dgrad_s2 if st.segs is None else lib.kodhip_conv_dgrad_s2_bnred
                chk(fn(st.raw.data_ptr(), dp + 2 * st.d_off, self._ptr(u.src, True),
                       B, st.H, st.W, u.src.buf.C, u.src.coff, u.cin, C_, C_, 0,
                       acc_src, fptr, *fz, s), u.name + ".dgrad")
            else:
                fn = lib.kodhip_conv_dgrad if st.segs is None else lib.kodhip_conv_dgrad_bnred
                chk(fn(st.raw.data_ptr(), dp + 2 * st.d_off, self._ptr(u.src, True),
                       *geo, st.Kdp, C_, 0, acc_src, fptr, *fz, s), u.name + ".dgrad")
            self._t1(e0, "dgrad" if st.segs is None else "dgrad+bn_reduce", nb)
        cin_true = 3 if u.stem else u.cin
        in_px_w = B * H * W if u.stem else B * st.H * st.W
        timed_wgrad(u.name, 2.0 * (in_px_w * cin_true + st.M * C_),
                    self._ptr(u.src), st.raw.data_ptr(), wgp + 4 * st.wg_off, gp + 4 * st.w_off,
                    *geo, st.Kp, C_, 0, C_, 1 if u.stem else 0, 1.0)
        self._flush_wgrads()           # this unit's - and a fused short_conv partner's - weight gradients: after the dgrad


    def _comm_stream(self):
        """Stream of the gradient-bucket all-reduces.  Default (comm_overlap): the WEIGHT-GRADIENT side stream, through
        the buckets' own communicator - a bucket is enqueued right behind the last weight gradient that fills it and
        overlaps the rest of backward on the main stream (torch DDP's reducer does the same with its hooks; north_star:
        "all-reduce overlapped with the backward pass").  SyncBN sums (main stream, `comm`) and buckets (side stream,
        `comm_buckets`) never share a communicator, so no communicator sees calls from two streams; every rank enqueues
        the same program, so the order inside each stream / graph branch is the same on all ranks.
        KODHIP_COMM_OVERLAP=0: None = everything on the main stream in one order (the conservative switch)."""
        if not self.comm_overlap or self.wg_stream is None or not self.wgrad_overlap:
            return None
        return self.wg_stream

    def wait_grads(self):
        for w in self._pending:
            w.wait()
        self._pending = []

    def _publish_grads(self):
        """Expose the arena slices as .grad (accumulating into an existing .grad like autograd would)."""
        cur = self.g_arena[self.g_cur]
        other = self.g_arena[self.g_cur ^ 1]
        first = next(iter(self.layout))
        existing = self.params[first].grad
        if existing is not None and existing.data_ptr() == self._grad_view(first, other).data_ptr():
            self.wait_grads()
            other.add_(cur)                      # gradient accumulation across backward() calls
            return
        for n in self.layout:
            p = self.params[n]
            if p.grad is not None and p.grad.data_ptr() != self._grad_view(n, cur).data_ptr():
                raise RuntimeError("mixed external .grad tensors are not supported; call zero_grad(set_to_none=True)")
            p.grad = self._grad_view(n, cur)
        self.g_cur ^= 1

    def current_grad_arena(self):
        """Arena holding the gradients published by the last backward()."""
        return self.g_arena[self.g_cur ^ 1]

    # ------------------------------------------------------------------ optimizer
    def set_hyper(self, lr, momentum, weight_decay, grad_scale: float = 1.0):
        """Upload the optimizer hyper-parameters (3-tuples for bias_params, decay_params, norm_params) to the device
        buffer the fused SGD kernel reads - outside any captured graph, so schedules keep working under replay."""
        vals = (*lr, *momentum, *weight_decay, grad_scale)
        if vals != self._hyper_vals:                       # only touch the device copy when the schedule moved
            k = self._hyper_slot
            self._hyper_slot = (k + 1) % len(self._hyper_host)
            if self._hyper_events[k] is not None:          # the DMA that last read this pinned slot must have run
                self._hyper_events[k].synchronize()
            host = self._hyper_host[k]
            host.copy_(torch.tensor(vals, dtype=torch.float32))
            self.hyper.copy_(host, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._hyper_events[k] = ev
            self._hyper_vals = vals

    def sgd_step(self, lr, momentum, weight_decay, grad_scale: float = 1.0):
        """lr / momentum / weight_decay: 3-tuples for (bias_params, decay_params, norm_params)."""
        self.wait_grads()
        self.set_hyper(lr, momentum, weight_decay, grad_scale)
        self.sgd_step_device()

    def sgd_step_device(self):
        """SGD with whatever is in self.hyper (device, 10 floats) - the graph-capturable form."""
        _lib.check(self.lib.kodhip_sgd_nesterov(self.p_arena.data_ptr(), self.current_grad_arena().data_ptr(),
                                                self.m_arena.data_ptr(), self.gid.data_ptr(), self.n_arena,
                                                self.hyper.data_ptr(), self._stream()), "sgd")
        self.param_version += 1

    def mark_params_changed(self):
        self.param_version += 1

    def invalidate_eval_constants(self):
        """Call after anything the version counters cannot see changed parameters or running statistics - i.e. a
        replay of a user-captured hipGraph that contains a training forward or an optimizer step (GraphedTrainStep
        does it itself)."""
        self.param_version += 1
        self.stats_version += 1
