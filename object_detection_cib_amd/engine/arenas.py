"""Parameter side of the engine: flat fp32 arenas (masters, gradients x 2, momentum), BatchNorm statistic arenas, the
bf16 MFMA operand packs and the fused SGD step.

Replaces what torch.optim.SGD's foreach kernels and autograd's .grad bookkeeping do for the reference
(kod/nn/optim/smart.py:36-58, kod/lightning/experiments/yv5_baseline/exp.py:156-185): torch Parameters are views of ONE
arena in forward execution order, so gradients complete back-to-front, data-parallel buckets are contiguous slices and
the optimizer is one launch.  Mixed into engine.executor.Engine.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import numpy as np

import torch

from .. import _lib
from .graph import Graph, ConvUnit, HeadUnit, View, Buf, head_param


def _pad(n: int, a: int = 64) -> int:
    return (n + a - 1) // a * a


class _UnitState:
    """Per conv unit: arena offsets (set once) and the current shape set's buffers / launch state."""
    __slots__ = ("u", "w_off", "g_off", "b_off", "f_off", "d_off", "Kp", "Kdp", "rs_off", "stats", "T",
                 "sums", "aff", "bsums", "bsums_g", "bpart", "T2", "coef", "raw", "M", "H", "W", "Ho", "Wo",
                 "fused_red", "segs", "seg_slots", "Kp_f", "raw_ld", "s2_fold", "wg_splits", "wg_off", "stem_fused", "wg_dual",
                 "pair", "pair_raw")


class ArenaMixin:
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
            place([head_param(h, k, "weight") for k in ("box", "obj", "cls")], 1)
            place([head_param(h, k, "bias") for k in ("box", "obj", "cls")], 0)
        self.n_arena = off
        self.layout = layout
        self.unit_starts = ([layout[u.name + '.0.weight'][0] for u in exec_units]
                            + [layout[head_param(h, 'box', 'weight')][0] for h in g.heads])
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
                      w_off=layout[head_param(h, "box", "weight")][0],
                      b_off=layout[head_param(h, "box", "bias")][0])
            n_off = 0
            for k, n in (("box", 4 * A), ("obj", A), ("cls", nc * A)):
                add_desc(head_param(h, k, "weight"), foff + n_off * Kp, doff, n, h.cin, 1, 1, Kp, Kdp,
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
        self.hyper = torch.zeros(12, dtype=torch.float32, device=device)      # lr[3] | momentum[3] | wd[3] | grad scale | flags | dampening
        self.sgd_nesterov = True          # FusedSGD(nesterov=...): smart_sgd.yaml's default, kod/configs/nn/optimizers/smart_sgd.yaml
        self.sgd_dampening, self.sgd_maximize = 0.0, False      # torch.optim.SGD(dampening=, maximize=) through FusedSGD
        self.sgd_steps = 0                # optimizer steps taken (torch's first step copies the gradient into the momentum buffer)
        self._hyper_args = None
        # pinned staging ring: the H2D copy is asynchronous, so a slot is not rewritten for the next 15 uploads
        self._hyper_host = [torch.zeros(12, dtype=torch.float32).pin_memory() for _ in range(16)]
        self._hyper_events = [None] * len(self._hyper_host)
        self._hyper_slot = 0
        self._hyper_vals = None

    def _grad_view(self, name, arena=None):
        o, k = self.layout[name]
        a = self.g_arena[self.g_cur] if arena is None else arena
        return a[o:o + k].view(self.params[name].shape)

    def pack_weights(self):
        _lib.check(self.lib.kodhip_pack_weights(self.p_arena.data_ptr(), self.fpack.data_ptr(),
                                                self.dpack.data_ptr(), self.pack_descs.data_ptr(),
                                                self.pack_descs.shape[0], self.pack_blocks, self._stream()),
                   "pack_weights")
        self._packed_version = self.param_version

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
        self._hyper_args = (tuple(lr), tuple(momentum), tuple(weight_decay), grad_scale)
        first = self.sgd_dampening != 0.0 and self.sgd_steps == 0
        flags = (1.0 if self.sgd_nesterov else 0.0) + (2.0 if self.sgd_maximize else 0.0) + (4.0 if first else 0.0)
        vals = (*lr, *momentum, *weight_decay, grad_scale, flags, float(self.sgd_dampening))
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
        """SGD with whatever is in self.hyper (device, 12 floats) - the graph-capturable form."""
        _lib.check(self.lib.kodhip_sgd_nesterov(self.p_arena.data_ptr(), self.current_grad_arena().data_ptr(),
                                                self.m_arena.data_ptr(), self.gid.data_ptr(), self.n_arena,
                                                self.hyper.data_ptr(), self._stream()), "sgd")
        self.param_version += 1
        self.note_sgd_step()

    def note_sgd_step(self):
        """one optimizer step has run (eagerly, or inside a replayed graph): with dampening the first step's flag must leave
        the device block before the next one"""
        self.sgd_steps += 1
        if self.sgd_dampening != 0.0 and self.sgd_steps == 1 and self._hyper_args is not None:
            self.set_hyper(*self._hyper_args)

    def mark_params_changed(self):
        self.param_version += 1

    def invalidate_eval_constants(self):
        """Call after anything the version counters cannot see changed parameters or running statistics - i.e. a
        replay of a user-captured hipGraph that contains a training forward or an optimizer step (GraphedTrainStep
        does it itself)."""
        self.param_version += 1
        self.stats_version += 1
