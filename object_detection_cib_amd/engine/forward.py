"""Forward launch program: walks the static op list (engine/graph.py) and enqueues libkodhip kernels on the caller's
stream (+ side streams for the CSP short_conv branches and the P3 / P4 heads).

Replaces what aten does for the reference's `net(images)` (kod/nn/networks/yolov5.py:90-108): conv -> train-mode
BatchNorm -> SiLU units, torch.cat / nn.Upsample as channel-slice writes, the three fused heads.  Mixed into Engine.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import numpy as np

import torch

from .. import _lib
from .graph import Graph, ConvUnit, HeadUnit, View, Buf

BN_EPS, BN_MOMENTUM = 1e-3, 0.03        # kod/nn/networks/yolov5.py:24


class ForwardMixin:
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

    def _t1(self, e0, family: str, nbytes: float, stream=None, name: str = ""):
        if e0 is None:
            return
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record(stream) if stream is not None else e1.record()
        self.profile.append((family, e0, e1, nbytes, name))          # name: the unit(s) the launch belongs to (tools/layer_table.py)

    def _stamp(self, name: str, stream=None):
        """(debug) device time stamp `name` on `stream` (a torch stream; default: the current one)"""
        if not self.stamps_on:
            return
        if self.stamp_buf is None:
            assert torch.device(self.device).type == "cuda"
            self.stamp_buf = torch.zeros(1024, dtype=torch.int64, device=self.device)
        assert self.stamp_buf.is_cuda and len(self.stamp_names) < 1000
        if name not in self.stamp_names:
            self.stamp_names.append(name)
        i = self.stamp_names.index(name)
        sid = stream.cuda_stream if stream is not None else self._stream()
        _lib.check(self.lib.kodhip_debug_stamp(self.stamp_buf.data_ptr() + 8 * i, sid), "stamp")

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
    def image_buffer(self, B: int, H: int, W: int) -> torch.Tensor:
        """The network's input buffer for this shape: bf16 pixel pairs [B, H, W/2, 8] (channels r, g, b, 0 of two
        neighbouring pixels).  Fill it and call forward(..., image_ready=True)."""
        assert not self.g.inputs
        self.allocate(B, H, W)
        return self.act["image"]

    # ------------------------------------------------------------------ forward
    def forward(self, x, training: bool = True, after_first_layer=None, image_ready: bool = False):
        """Whole network / backbone: x = [B,3,H,W] fp32 NCHW on this device.  Sub-network graphs (Graph.inputs): x = a
        sequence of NCHW tensors, one per input view (copied into the channels-last bf16 buffers; torch does the layout
        change, the arithmetic stays in libkodhip).  Returns the head tensors [B,A,h,w,5+nc] fp32 (ll, ml, hl) followed
        by the Graph.outputs views as NCHW fp32 tensors.
        after_first_layer: called once after the first layer's kernels are launched (a hook for side-stream work that only
        depends on the step's inputs: it is then captured behind the forward chain's head, see Yolov5Network.train_step).
        image_ready: the caller has already put the batch into the engine's own input buffer (`image_buffer()`: bf16 pixel
        pairs [B, H, W/2, 8], what kodhip_nchw_to_nhwc4 produces and kodhip_compose_batch can write directly) - x then only
        carries the shape, and the layout-change pass is skipped (the training loop of bench.py / DeviceTrainPipeline)."""
        lib, chk = self.lib, _lib.check
        if self.g.inputs:
            xs = list(x)
            assert len(xs) == len(self.g.inputs)
            v0 = self.g.inputs[0]
            B, H, W = xs[0].shape[0], xs[0].shape[2] * v0.stride, xs[0].shape[3] * v0.stride
            self.allocate(B, H, W)
            for v, t in zip(self.g.inputs, xs):
                assert tuple(t.shape) == (B, v.C, H // v.stride, W // v.stride) and t.device == self.device, (t.shape, v.C, v.stride)
                self.act[v.buf.name][..., v.coff:v.coff + v.C].copy_(t.permute(0, 2, 3, 1))
        else:
            B, Cimg, H, W = x.shape
            assert Cimg == 3 and x.dtype == torch.float32 and x.is_contiguous() and x.device == self.device
            self.allocate(B, H, W)
        s = self._stream()
        if self._packed_version != self.param_version:
            # (round 6: the re-pack on a side stream beside the input's layout change measured 1.3 % SLOWER in the replayed
            # step - a second root node moves the forward chain's head to another hardware queue; LOG round 6)
            self.pack_weights()
        if not self.g.inputs and not image_ready:
            chk(lib.kodhip_nchw_to_nhwc4(x.data_ptr(), self.act["image"].data_ptr(), B, 3, H, W, s), "nchw_to_nhwc4")
        self._stamp("fwd_begin")
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
                                        st.stats.data_ptr(), *geo, st.raw_ld, 0, s), u.name)
            cin_true = 3 if u.stem else u.cin
            in_px = B * H * W if u.stem else B * st.H * st.W
            self._t1(e0, "conv_fwd", 2 * (in_px * cin_true + st.M * C_), name=u.name)

        def pair_stage(mu: ConvUnit, su: ConvUnit, s=s):
            """A CSP layer's main_conv + short_conv (same input): one convolution with N = 2 * mid columns, one launch for
            both units' BatchNorm constants, one apply pass writing each half to its own destination slice."""
            mst, sst, mid = self.ustate[mu.name], self.ustate[su.name], mu.cout
            geo = (B, mst.H, mst.W, mu.src.buf.C, mu.src.coff, mu.cin, 2 * mid, 1, 1, 1, 1, 0, 0, mst.Kp_f)
            e0 = self._t0()
            chk(lib.kodhip_conv_fwd_raw(self._ptr(mu.src), fp + 2 * mst.f_off, mst.pair_raw.data_ptr(),
                                        mst.stats.data_ptr(), *geo, 2 * mid, 0, s), mu.name + "+short")
            self._t1(e0, "conv_fwd", 2 * (B * mst.H * mst.W * mu.cin + mst.M * 2 * mid))
            ma, sa = mst.aff.data_ptr(), sst.aff.data_ptr()
            if training:
                e0 = self._t0()
                peer = sync and self.peer is not None
                chk(lib.kodhip_bn_finalize_partials_pair(
                    mst.stats.data_ptr(), mst.T, float(mst.M) * (self.world_size if sync else 1), mid, self.bn_momentum, self.bn_eps, 1,
                    pa + 4 * mst.g_off, pa + 4 * mst.b_off, rm + 4 * mst.rs_off, rv + 4 * mst.rs_off, ma,
                    pa + 4 * sst.g_off, pa + 4 * sst.b_off, rm + 4 * sst.rs_off, rv + 4 * sst.rs_off, sa,
                    self.peer.view_ptr() if peer else None,
                    self.peer_slots[(mu.name, "f")] if peer else 0, self.peer_slots[(su.name, "f")] if peer else 0, s),
                    mu.name + "+short")
                self._t1(e0, "bn_finalize", 8.0 * 2 * mid * mst.T)
            (msc, msh), (ssc, ssh) = ((ma, ma + 4 * mid), (sa, sa + 4 * mid)) if training else (eval_aff[mu.name], eval_aff[su.name])
            if self.opt.pair_fwd == 2 and branch:
                # the short half feeds only last_conv: its apply pass leaves the main chain (side stream, joined there)
                if self.br_stream is None:
                    self.br_stream = torch.cuda.Stream(device=self.device)
                fork = torch.cuda.Event()
                fork.record(main_stream)
                e0 = self._t0()
                chk(lib.kodhip_bn_silu_apply(mst.raw.data_ptr(), 2 * mid, msc, msh, None, 0, 0,
                                             self._ptr(mu.dst), mu.dst.buf.C, mu.dst.coff, mst.M, mid, s), mu.name)
                self._t1(e0, "bn_silu_apply", 4.0 * mst.M * mid)
                self.br_stream.wait_event(fork)
                chk(lib.kodhip_bn_silu_apply(sst.raw.data_ptr(), 2 * mid, ssc, ssh, None, 0, 0,
                                             self._ptr(su.dst), su.dst.buf.C, su.dst.coff, mst.M, mid,
                                             self.br_stream.cuda_stream), su.name)
                return su.dst.buf.name
            e0 = self._t0()
            chk(lib.kodhip_bn_silu_apply_pair(mst.pair_raw.data_ptr(), 2 * mid, mid,
                                              msc, msh, self._ptr(mu.dst), mu.dst.buf.C, mu.dst.coff,
                                              ssc, ssh, self._ptr(su.dst), su.dst.buf.C, su.dst.coff, mst.M, s), mu.name + "+short")
            self._t1(e0, "bn_silu_apply", 4.0 * mst.M * 2 * mid)
            return None

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
                                                        self.bn_momentum, self.bn_eps, aff, aff + 4 * C_, aff + 8 * C_,
                                                        aff + 12 * C_, C_, 1, s), u.name)
            elif self.peer is not None:
                # SyncBN over peer buffers: the same single launch per unit, the ranks' sums meet inside the kernel
                for u in group:
                    st, C_ = self.ustate[u.name], u.cout
                    aff = st.aff.data_ptr()
                    chk(lib.kodhip_bn_finalize_partials_peer(st.stats.data_ptr(), st.T, float(st.M) * self.world_size,
                                                             pa + 4 * st.g_off, pa + 4 * st.b_off, rm + 4 * st.rs_off,
                                                             rv + 4 * st.rs_off, self.bn_momentum, self.bn_eps, aff, aff + 4 * C_,
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
                                               pa + 4 * st.b_off, rm + 4 * st.rs_off, rv + 4 * st.rs_off, self.bn_momentum,
                                               self.bn_eps, aff, aff + 4 * C_, aff + 8 * C_, aff + 12 * C_, C_, 1, s), u.name)
            self._t1(e0, "bn_finalize", sum(8.0 * u.cout * self.ustate[u.name].T for u in group), name="+".join(u.name for u in group))

        def apply_stage(u: ConvUnit, s=s):
            st, C_ = self.ustate[u.name], u.cout
            aff = st.aff.data_ptr()
            sc_p, sh_p = (aff, aff + 4 * C_) if training else eval_aff[u.name]
            res = u.residual
            e0 = self._t0()
            chk(lib.kodhip_bn_act_apply(st.raw.data_ptr(), st.raw_ld, sc_p, sh_p,
                                        self._ptr(res) if res else None, res.buf.C if res else 0,
                                        res.coff if res else 0,
                                        self._ptr(u.dst), u.dst.buf.C, u.dst.coff, st.M, C_, self.act_kind, self.act_slope, s), u.name)
            self._t1(e0, "bn_silu_apply", (6.0 if res else 4.0) * st.M * C_, name=u.name)

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
            if op.kind == "conv" and op.unit.sibling is not None and i < len(ops) and ops[i].unit is op.unit.sibling and \
                    self.ustate[op.unit.name].pair is not None and not (sync and self.peer is None):
                # (RCCL SyncBN keeps the two-launch form: its statistic exchange works on per-unit sum vectors)
                if joined_buf is not None:
                    main_stream.wait_stream(self.br_stream)
                joined_buf = pair_stage(op.unit, ops[i].unit)
                i += 1
                continue
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
                chk(lib.kodhip_maxpool_fwd(self._ptr(op.src), op.src.buf.C, op.src.coff, self._ptr(op.dst),
                                           op.dst.buf.C, op.dst.coff, self.pool_idx[pool_i].data_ptr(),
                                           B, h, w, op.src.C, op.k, s), "maxpool")
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
        self._stamp("fwd_end")
        if training:
            self.nbt_arena += 1
            self.stats_version += 1              # running statistics moved
        self.training_ready = training          # an eval forward overwrites the saved pre-BN tensors
        for v in self.g.outputs:                # sub-network graphs: their output views, NCHW fp32
            outs.append(self.act[v.buf.name][..., v.coff:v.coff + v.C].permute(0, 3, 1, 2).float())
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
            sc = self.p_arena[gi] * torch.rsqrt(self.rv_arena[ri] + self.bn_eps)
            self._eval_aff[:n] = sc
            self._eval_aff[n:] = self.p_arena[bi] - self.rm_arena[ri] * sc
            self._eval_key = key
        base, n = self._eval_aff.data_ptr(), self._eval_n
        return {name: (base + 4 * o, base + 4 * (n + o)) for name, o in self._eval_off.items()}
