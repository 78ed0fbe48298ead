"""Backward launch program: the forward op list in reverse - BatchNorm/SiLU backward, data gradients on the main
stream, weight gradients (+ gradient buckets) on a side stream, captured so that the critical chain stays on one
queue of the replayed hipGraph.

Replaces autograd's traversal for the reference's `total.backward()` (kod/lightning/experiments/yv5_baseline/
exp.py:104-138) and torch DDP's reducer hooks (kod/configs/trainer/ddp.yaml:4-9).  Mixed into Engine.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import numpy as np

import torch

from .. import _lib
from .graph import Graph, ConvUnit, HeadUnit, View, Buf, head_param
from .ddp import plan_buckets, launch_bucket


def _cu_masked_stream(device):
    """Probe (LOG round 6): KODHIP_WG_CUMASK=<hex word>[:<words>] confines the weight-gradient stream
    to a CU subset (hipExtStreamCreateWithCUMask; the 32-bit word is repeated over the chip's 256 CU bits).  None = unset."""
    import os
    spec = os.environ.get("KODHIP_WG_CUMASK")
    if not spec:
        return None
    import ctypes
    word, _, n = spec.partition(":")
    words = int(n) if n else 8
    mask = (ctypes.c_uint32 * words)(*([int(word, 16)] * words))
    hip = ctypes.CDLL("libamdhip64.so")
    st = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), mask)
    if rc != 0 or not st.value:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed ({rc})")
    return torch.cuda.ExternalStream(st.value, device=device)


class BackwardMixin:
    # ------------------------------------------------------------------ backward
    def backward(self, head_grads: List[torch.Tensor], out_grads: Optional[List[torch.Tensor]] = None):
        """head_grads: d loss / d (ll, ml, hl) head tensors.  Fills the gradient arena.  Sub-network graphs: out_grads =
        d loss / d Graph.outputs (NCHW, None = zero); returns d loss / d Graph.inputs (NCHW fp32), else nothing."""
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
                self.wg_stream = _cu_masked_stream(self.device) or torch.cuda.Stream(device=self.device)
                self.wg_more = [torch.cuda.Stream(device=self.device) for _ in range(self._wg_streams - 1)]
            wg = self.wg_stream
        # further weight-gradient streams (EngineOptions.wgrad_streams): the launches rotate over them, each stream with its
        # own slab scratch; stream 0 is the one the gradient buckets ride on
        wgs = [wg] + (self.wg_more if (wg is not None and self.wgrad_fork != "legacy") else []) if wg is not None else []
        rr = [0]

        def join_main():
            """the weight-gradient stream (where the gradient buckets ride) waits for what the main stream has queued"""
            if wg is not None and buckets:
                ev = torch.cuda.Event()
                ev.record(main)
                wg.wait_event(ev)
        self._join_main = join_main

        def join_wg():
            """stream 0 waits for everything queued on the other weight-gradient streams"""
            for o in wgs[1:]:
                ev = torch.cuda.Event()
                ev.record(o)
                wg.wait_event(ev)

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
            """args = kodhip_conv_wgrad's (x, dy, slab region, grad, geometry ..., n_valid, stem, scale), or
            ("stem", region offset in bytes, kodhip_stem_bwd_fused's arguments with the region base in place 9)"""
            e0 = self._t0(stream_obj)
            sid = stream_obj.cuda_stream if stream_obj is not None else s
            self._stamp("wg:" + name, stream_obj)
            if args[0] == "stem":
                fa = list(args[2:])
                fa[9] += args[1]
                chk(lib.kodhip_stem_bwd_fused(*fa, sid), name + ".bwd_fused")
            elif args[0] == "dual":           # ("dual", region offset, kodhip_conv_wgrad_dual's arguments)
                fa = list(args[2:])
                fa[3] += args[1]
                chk(lib.kodhip_conv_wgrad_dual(*fa, sid), name + ".wgrad2")
            elif batched:
                chk(lib.kodhip_conv_wgrad_partial(*args[:3], *args[4:-3], sid), name + ".wgrad")
            else:
                chk(lib.kodhip_conv_wgrad(*args, sid), name + ".wgrad")
            self._t1(e0, "wgrad", nbytes, stream_obj, name=name)

        def flush_wgrads():
            """call after the main stream's next kernel has been launched"""
            for ev, name, nbytes, args in deferred:
                k = rr[0] % len(wgs)
                rr[0] += 1
                if k:
                    args = list(args)
                    args[1 if isinstance(args[0], str) else 2] += 4 * k * self._wg_region
                wgs[k].wait_event(ev)
                launch_wgrad(name, nbytes, args, wgs[k])
            deferred.clear()
            if due and not hold[0]:
                self._launch_due()
        self._flush_wgrads = flush_wgrads
        # a short_conv whose weight gradient waits for its main_conv's dual launch: no gradient bucket may be enqueued meanwhile
        hold = [False]
        self._wg_hold = hold

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

        # sub-network graphs: the callers' output gradients are the first writers of those buffers
        if self.g.outputs:
            og = list(out_grads) if out_grads is not None else [None] * len(self.g.outputs)
            for v, t in zip(self.g.outputs, og):
                name = v.buf.name
                if name not in touched:
                    touched.add(name)
                    if v.C != v.buf.C or t is None:
                        self.gact[name].zero_()
                if t is not None:
                    self.gact[name][..., v.coff:v.coff + v.C].copy_(t.permute(0, 2, 3, 1))
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
                    join_wg()
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
                if not deferred and not hold[0]:
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
                    chk(lib.kodhip_bn_act_bwd_reduce(self._ptr(dA, True), dA.buf.C, dA.coff, st.raw.data_ptr(), st.raw_ld,
                                                     aff, aff + 4 * C_, aff + 8 * C_, aff + 12 * C_,
                                                     st.bpart.data_ptr(), st.M, C_, self.act_kind, self.act_slope, s), u.name)
                    self._t1(e0, "bn_bwd_reduce", 4.0 * st.M * C_, name=u.name)
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
                self._t1(e0, "bn_bwd_coeffs", sum(8.0 * u.cout * self.ustate[u.name].T2 for u in group), name="+".join(u.name for u in group))
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
            self._t1(e0, "bn_bwd_coeffs", sum(8.0 * u.cout * self.ustate[u.name].T2 for u in group), name="+".join(u.name for u in group))

        # (with every collective on the main stream - KODHIP_COMM_OVERLAP=0, RCCL SyncBN - the head chains stay there too)
        heads_side = (wg is not None and defer and self.branch_overlap and self.profile is None and
                      (not self.collectives or (self._comm_stream() is not None and not rccl_sync)))
        self._stamp("bwd_begin")
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
                names = [head_param(hu, k, "bias") for k in ("box", "obj", "cls")]
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
                self._t1(e0, "dgrad", 2.0 * hs["M"] * (self.head_npad + hu.cin), name=hu.name)
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
                chk(lib.kodhip_maxpool_bwd(self._ptr(op.dst, True), op.dst.buf.C, op.dst.coff,
                                           self.pool_idx[pool_i].data_ptr(), self._ptr(op.src, True),
                                           op.src.buf.C, op.src.coff, B, h, w, op.src.C, op.k, f32("pool", op, op.src)[1], s), "maxpool_bwd")
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
                                   partner=group[0] if dual else None,
                                   dual_w=dual and self.ustate[group[1].name].wg_dual > 0)
                    bucket_tick()
                continue
            # gradient buckets complete from the arena's end toward its start
            if op.kind == "head":
                bucket_tick()
        flush_wgrads()
        self._stamp("main_end")
        if wg is not None:
            self._stamp("wg_end", wg)
        for name in list(grad_events):
            sync_grad(name)
        for o in wgs:
            main.wait_stream(o)
        self._stamp("bwd_end")
        self._publish_grads()
        if self.g.inputs:
            return [self.gact[v.buf.name][..., v.coff:v.coff + v.C].permute(0, 3, 1, 2).float() if v.buf.name in touched
                    else torch.zeros((B, v.C, H // v.stride, W // v.stride), device=self.device) for v in self.g.inputs]

    def _bwd_unit(self, u, B, H, W, s, gp, pa, dp, wgp, acc_flag, timed_wgrad, dgrad="own", partner=None, dual_w=False):
        """bn/silu backward apply -> data gradient -> weight gradient of one conv unit (coefficients already in st.coef).
        dgrad: "own" = this unit's launch; "skip" = none (a fused short_conv: its main_conv's launch covers it);
        "dual" = one launch for this unit and `partner` (kodhip_conv_dgrad_dual)."""
        lib, chk = self.lib, _lib.check
        st = self.ustate[u.name]
        C_ = u.cout
        aff = st.aff.data_ptr()
        dA = u.dst
        res = u.residual
        self._stamp("m:" + u.name)
        if st.stem_fused and res is None:
            # the stem has no data gradient: dY = f(dA, y) is formed inside its weight gradient and never written
            # (csrc/conv_wgrad.hip conv_stem_bwd_fused_kernel); the launch joins the weight-gradient stream behind the
            # coefficient kernel
            fargs = (self._ptr(u.src), self._ptr(dA, True), dA.buf.C, dA.coff, st.raw.data_ptr(), st.raw_ld,
                     aff, aff + 4 * C_, st.coef.data_ptr())
            nb = 2.0 * (B * H * W * 3 + 2 * st.M * C_)
            if self.opt.native.get("KODHIP_STEM_BWD_STREAM", "main") == "main":
                # on the MAIN stream, with a slab scratch of its own: it is the main chain's last kernel, and the chip is
                # otherwise left to the tail of the weight-gradient stream (small launches, one at a time) - this HBM-bound
                # kernel runs beside them instead of behind them
                e0 = self._t0()
                chk(lib.kodhip_stem_bwd_fused(*fargs, self.stem_part.data_ptr(), gp + 4 * st.w_off,
                                              B, st.H, st.W, C_, 1.0, s), u.name + ".bwd_fused")
                self._t1(e0, "wgrad", nb, name=u.name)
                self._flush_wgrads()
                self._join_main()          # a gradient bucket on the weight-gradient stream must see this gradient
                return
            self._fork_point()
            timed_wgrad(u.name, nb, "stem", 0, *fargs, wgp + 4 * st.wg_off, gp + 4 * st.w_off, B, st.H, st.W, C_, 1.0)
            self._flush_wgrads()
            return
        racc = acc_flag(res) if res else 0
        e0 = self._t0()
        chk(lib.kodhip_bn_act_bwd_apply(self._ptr(dA, True), dA.buf.C, dA.coff, st.raw.data_ptr(), st.raw_ld,
                                        aff, aff + 4 * C_, st.coef.data_ptr(),
                                        self._ptr(res, True) if res else None,
                                        res.buf.C if res else 0, res.coff if res else 0,
                                        racc, st.M, C_, self.act_kind, self.act_slope, s), u.name)
        self._t1(e0, "bn_silu_bwd_apply", (6.0 + ((4.0 if racc else 2.0) if res else 0.0)) * st.M * C_, name=u.name)
        self._fork_point()
        # st.raw now holds dY
        if u.stem:
            geo = (B, st.H, st.W, 8, 0, 8, C_, 6, 3, 2, 1, 2, 1)
        else:
            geo = (B, st.H, st.W, u.src.buf.C, u.src.coff, u.cin, C_, u.k, u.k, u.s, u.s, u.p, u.p)
            fz = () if st.segs is None else (C.cast(st.segs, C.c_void_p), len(st.segs), st.seg_slots)
            if dgrad == "skip" and dual_w:          # its weight gradient rides in the main_conv's dual launch
                self._wg_hold[0] = True
                return
            if dgrad == "skip":
                timed_wgrad(u.name, 2.0 * (B * st.H * st.W * u.cin + st.M * C_),
                            self._ptr(u.src), st.raw.data_ptr(), wgp + 4 * st.wg_off, gp + 4 * st.w_off,
                            *geo, st.Kp, st.raw_ld, 0, C_, 0, 1.0)
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
                       B, st.H, st.W, u.src.buf.C, u.src.coff, u.cin, C_, st.Kdp, st.raw_ld, 0, acc_src, fptr, *fz, s), u.name + ".dgrad2")
            elif u.k == 3 and u.s == 2 and u.p == 1:
                if st.s2_fold:
                    fn = lib.kodhip_conv_dgrad_s2f if st.segs is None else lib.kodhip_conv_dgrad_s2f_bnred
                else:
                    fn = lib.kodhip_conv_dgrad_s2 if st.segs is None else lib.kodhip_conv_dgrad_s2_bnred
                chk(fn(st.raw.data_ptr(), dp + 2 * st.d_off, self._ptr(u.src, True),
                       B, st.H, st.W, u.src.buf.C, u.src.coff, u.cin, C_, st.raw_ld, 0,
                       acc_src, fptr, *fz, s), u.name + ".dgrad")
            else:
                fn = lib.kodhip_conv_dgrad if st.segs is None else lib.kodhip_conv_dgrad_bnred
                chk(fn(st.raw.data_ptr(), dp + 2 * st.d_off, self._ptr(u.src, True),
                       *geo, st.Kdp, st.raw_ld, 0, acc_src, fptr, *fz, s), u.name + ".dgrad")
            self._t1(e0, "dgrad" if st.segs is None else "dgrad+bn_reduce", nb, name=u.name + ("+" + partner.name if dgrad == "dual" else ""))
        if dgrad == "dual" and dual_w:
            ps = self.ustate[partner.name]
            self._wg_hold[0] = False
            timed_wgrad(u.name + "+" + partner.name, 2.0 * (B * st.H * st.W * u.cin + 2 * st.M * C_),
                        "dual", 0, self._ptr(u.src), st.raw.data_ptr(), ps.raw.data_ptr(), wgp + 4 * st.wg_off,
                        gp + 4 * st.w_off, gp + 4 * ps.w_off, B, st.H, st.W, u.src.buf.C, u.src.coff, u.cin, C_, st.Kp,
                        st.raw_ld, 0, 1.0)
            self._flush_wgrads()
            return
        cin_true = 3 if u.stem else u.cin
        in_px_w = B * H * W if u.stem else B * st.H * st.W
        timed_wgrad(u.name, 2.0 * (in_px_w * cin_true + st.M * C_),
                    self._ptr(u.src), st.raw.data_ptr(), wgp + 4 * st.wg_off, gp + 4 * st.w_off,
                    *geo, st.Kp, st.raw_ld, 0, C_, 1 if u.stem else 0, 1.0)
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
