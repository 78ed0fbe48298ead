"""Per-shape buffer sets of the engine (activations, activation gradients, pre-BN tensors, statistic slots, weight-
gradient slabs, fp32 shadows) and the shape-dependent launch plans that hold pointers into them.

A captured hipGraph bakes buffer addresses in, so every (B, H, W) keeps its own complete set and a forward at another
shape swaps pointers instead of reallocating.  The decisions themselves (who writes which gradient buffer last, which
gradients need fp32 accumulation, bucket boundaries) are pure functions in engine/plan.py and engine/ddp.py; this
module asks libkodhip for the slot / split counts of each launch and allocates accordingly.  Mixed into Engine.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import numpy as np

import torch

from .. import _lib
from .graph import Graph, ConvUnit, HeadUnit, View, Buf
from .arenas import _pad
from .ddp import plan_buckets
from .plan import backward_writes, plan_f32_accumulation, plan_dual_dgrads, plan_bn_reduce_fusion


class BufferMixin:
    # ------------------------------------------------------------------ activations
    _UNIT_FIELDS = ("stats", "T", "sums", "aff", "bsums", "bsums_g", "bpart", "T2", "coef", "raw", "M", "H", "W", "Ho",
                    "Wo", "fused_red", "segs", "seg_slots", "raw_ld", "wg_splits", "wg_off", "stem_fused", "wg_dual", "pair", "pair_raw")
    _HEAD_FIELDS = ("H", "W", "M", "dy", "ws", "wg_splits", "wg_off")

    def _export_set(self) -> dict:
        return dict(act=self.act, gact=self.gact, gact32=self.gact32, wg_part=self.wg_part, pool_idx=self.pool_idx,
                    red_groups=self.red_groups, wg_region=self._wg_region, stem_part=getattr(self, "stem_part", None),
                    units={n: {f: getattr(st, f) for f in self._UNIT_FIELDS} for n, st in self.ustate.items()},
                    heads={n: {f: hs[f] for f in self._HEAD_FIELDS} for n, hs in self.hstate.items()})

    def _import_set(self, d: dict):
        self.act, self.gact, self.wg_part, self.pool_idx = d["act"], d["gact"], d["wg_part"], d["pool_idx"]
        self.gact32, self.red_groups, self._wg_region = d["gact32"], d["red_groups"], d["wg_region"]
        self.stem_part = d["stem_part"]
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
        top = max(b.stride for b in self.g.bufs)
        assert H % top == 0 and W % top == 0, f"input size must be a multiple of {top} (the graph's coarsest map)"
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
        own = self.opt.wgrad_reduce_batched        # slab regions: one per layer (batched reduction) or one shared scratch
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
            nslab = st.wg_splits * u.cout * st.Kp
            # the stem's backward as one kernel (kodhip_stem_bwd_fused): a slab per block, 32 (cout <= 32) or 64 rows
            st.stem_fused = bool(u.stem and u.cout <= 64 and self.opt.stem_bwd_fused and not own)
            if st.stem_fused:
                st.wg_splits = lib.kodhip_stem_bwd_fused_blocks(B, st.H, st.W, u.cout)
                nslab = st.wg_splits * (32 if u.cout <= 32 else 64) * 160
                self.stem_part = torch.empty(nslab, dtype=torch.float32, device=dev)    # (it runs on the main stream)
            # slab region [splits][cout][Kp] (floats): ONE scratch shared by all layers (reduced right after each weight
            # gradient, while it is still in the 256 MB Infinity Cache) - or, for the per-bucket reduction, a region each
            st.wg_off = max_part if own else 0
            max_part = max_part + _pad(nslab) if own else max(max_part, nslab)
        # A CSP layer's main_conv and short_conv (kod/nn/layers/csp.py:87-88: the same input through two pointwise convs)
        # run forward as ONE convolution with N = 2 * mid columns: their packed weights are adjacent (arena order = forward
        # order), their pre-BN outputs are the two channel halves of one tensor (row stride 2 * mid: every later kernel takes
        # the half as a (pointer, row stride) slice) and their statistic slots are one [2][2 * mid][T] block.
        units_by_name = {u.name: u for u in self.exec_units}
        for u in self.exec_units:
            self.ustate[u.name].pair, self.ustate[u.name].pair_raw = None, None
        if self.opt.pair_fwd:
            from .plan import plan_dual_dgrads as _pairs
            for mname, sname in _pairs(self.g).items():
                mu, su = units_by_name[mname], units_by_name[sname]
                mst, sst = self.ustate[mname], self.ustate[sname]
                if mu.residual is not None or su.residual is not None or sst.f_off != mst.f_off + mu.cout * mst.Kp_f:
                    continue
                mid = mu.cout
                # + a zeroed tail that nobody writes: a data gradient whose channel count is not a multiple of 32 reads up to 16
                # channels past its slice against zero weights (padded-tap K axis); for the short_conv
                # half of the LAST row that is past the tensor, and the launcher's buffer range (rows x row stride from the
                # half's own base pointer) no longer ends where the allocation does
                flat = torch.empty(mst.M * 2 * mid + 512, dtype=torch.bfloat16, device=dev)
                flat[mst.M * 2 * mid:].zero_()
                both = flat[:mst.M * 2 * mid].view(B, mst.Ho, mst.Wo, 2 * mid)
                mst.pair_raw = sst.pair_raw = both
                mst.raw, sst.raw = both[..., :mid], both[..., mid:]
                mst.raw_ld = sst.raw_ld = 2 * mid
                mst.T = sst.T = lib.kodhip_conv_stats_slots(mst.M, 2 * mid)
                # (the short_conv keeps its own slots for the two-launch form: RCCL SyncBN, KODHIP_NO_PAIR_FWD)
                mst.stats = torch.empty(2 * 2 * mid * mst.T, dtype=torch.float32, device=dev)
                mst.pair, sst.pair = ("main", sname), ("short", mname)
                for st_, u_ in ((mst, mu), (sst, su)):          # (the row stride enters the launcher's 32-bit range test)
                    st_.wg_splits = lib.kodhip_conv_wgrad_splits_geo(B, st_.H, st_.W, u_.src.buf.C, u_.cin, u_.cout, 1, 1, 1, 1, 0, 0,
                                                                     st_.Kp, st_.raw_ld)
                    if not own:
                        max_part = max(max_part, st_.wg_splits * u_.cout * st_.Kp)
        self._plan_bn_fusion(B)
        # a dual pair's weight gradients as one launch (kodhip_conv_wgrad_dual): slab rows for both layers
        for u in self.exec_units:
            self.ustate[u.name].wg_dual = 0
        if self.opt.dual_wgrad and not own:
            for mname in self._dual:
                st = self.ustate[mname]
                u = st.u
                st.wg_dual = lib.kodhip_conv_wgrad_dual_splits(B, st.H, st.W, u.src.buf.C, u.cin, u.cout, st.Kp, st.raw_ld)
                max_part = max(max_part, st.wg_dual * 2 * u.cout * st.Kp)
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
        # shared scratch: one region per weight-gradient stream (engine/backward.py)
        self._wg_region = _pad(max_part)
        self.wg_part = torch.empty(max_part if own else self._wg_region * self._wg_regions, dtype=torch.float32, device=dev)
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
        """Turns the static plans of engine/plan.py into launch state for this shape: which CSP entry convs share a dual
        data gradient, which multi-producer gradients accumulate in fp32 (EngineOptions.dx_accum_fp32), and which data
        gradients carry the BatchNorm-backward reduction of the units whose output gradient they complete
        (kodhip_conv_dgrad_*_bnred: segment tables + partial buffers sized by the library's slot queries)."""
        lib = self.lib
        units = {u.name: u for u in self.exec_units}
        for u in self.exec_units:
            st = self.ustate[u.name]
            st.fused_red, st.segs, st.seg_slots = False, None, 0
        # dual data gradients: a CSP layer's main_conv and short_conv (both pointwise, same input) write dX in ONE launch
        self._dual = {m: units[sh] for m, sh in plan_dual_dgrads(self.g).items()} if self.opt.dual_dgrad else {}
        dual_shorts = {v.name for v in self._dual.values()}
        ws, upos = backward_writes(self.g, dual_shorts)          # who writes which gradient buffer, in backward order
        # activation gradients with several producers: accumulated in fp32 (one rounding) instead of bf16 read-modify-write
        self._f32plan = None
        if self.opt.dx_accum_fp32:
            self._f32plan = plan_f32_accumulation(ws, {b.name: b.C for b in self.g.bufs})
            if self.opt.debug_plan:
                print(f"[kodhip] fp32 accumulation of multi-producer gradients: shadows {sorted(self._f32plan.shadow_bufs)}; "
                      f"bf16 (unsupported) {self._f32plan.unsupported}", flush=True)
        if not self.opt.bn_reduce_fused:
            return
        for wname, prods in plan_bn_reduce_fusion(self.g, ws, upos).items():
            w = units[wname]
            wst = self.ustate[wname]
            s2 = int(w.k == 3 and w.s == 2 and w.p == 1)
            # (A/B knob: only fuse into launches whose reduction length is at least KODHIP_BNRED_MINK; measured best: all)
            if w.k * w.k * w.cout < self.opt.bn_reduce_min_k:
                continue
            if wname in self._dual:
                slots = lib.kodhip_conv_dgrad_dual_bnred_slots(B, wst.H, wst.W, w.cin, w.cout, wst.raw_ld)
            elif s2 and wst.s2_fold:
                slots = lib.kodhip_conv_dgrad_s2f_bnred_slots(B, wst.H, wst.W, w.cin, w.cout, w.cout)
            else:
                slots = lib.kodhip_conv_dgrad_bnred_slots(B, wst.H, wst.W, w.cin, w.cout, w.k, w.k, w.s, w.s, w.p, w.p, wst.raw_ld, s2)
            if slots <= 0:
                continue
            segs = (_lib.KodBnRedSeg * len(prods))()
            for i, (pname, ch0) in enumerate(prods):
                u, st = units[pname], self.ustate[pname]
                st.fused_red, st.T2 = True, slots
                st.bpart = torch.empty(2 * u.cout * slots, dtype=torch.float32, device=self.device)
                segs[i].ch_begin, segs[i].ch_count = ch0, u.cout
                segs[i].raw, segs[i].ldr = st.raw.data_ptr(), st.raw_ld
                segs[i].aff, segs[i].partials = st.aff.data_ptr(), st.bpart.data_ptr()
            wst.segs, wst.seg_slots = segs, slots
        if self.opt.debug_plan:
            fused = [u.name for u in self.exec_units if self.ustate[u.name].fused_red]
            print(f"[kodhip] BN-backward reduction fused into a data gradient for {len(fused)} of {len(self.exec_units)} units; "
                  f"separate pass: {[u.name for u in self.exec_units if not self.ustate[u.name].fused_red]}", flush=True)
