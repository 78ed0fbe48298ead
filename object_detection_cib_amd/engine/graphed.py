"""GraphedTrainStep - one training step (forward, assigner + loss, backward, gradient all-reduce, fused SGD) captured
once as a hipGraph and replayed on new data.

The reference leaves launch scheduling to PyTorch / Lightning (one aten kernel at a time from Python).  Here the
step is ~520 launches of 5-100 us; issued from Python it is host-bound, so the capture is what makes a training LOOP
run at device speed (bench.py measures exactly this replay).  What makes the step capturable:

* inputs live in static device buffers that new batches are copied into: the image batch, and the boxes / labels /
  image indices of all targets padded to a fixed capacity with zero-size boxes, which the assigner never matches
  (kod/core/label_assignment/yv5.py:262-296: the anchor-ratio test rejects w = h = 0);
* optimizer hyper-parameters are read from device memory (Engine.set_hyper), so warm-up / LR schedules keep working;
* no host synchronisation anywhere in the step (assigner counts stay on the device).
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from .. import _lib
from ..core.label_assignment.yv5 import BatchedTargets
from ..core.types import FeatureShape


class GraphedTrainStep:
    def __init__(self, net, loss, batch_size: int, height: int, width: int, max_targets: int = 4096,
                 loss_scale: Optional[float] = None, input_pairs: bool = False):
        """input_pairs: batches arrive as bf16 pixel pairs [B, H, W/2, 8] (DeviceTrainPipeline.make_batch(out_pairs=True)) and
        are copied straight into the network's input buffer - no fp32 NCHW batch, no layout-change pass in the step."""
        self.net, self.loss = net, loss
        self.input_pairs = bool(input_pairs)
        self.eng = net.engine()
        dev = self.eng.device
        self.B, self.H, self.W = batch_size, height, width
        self.cap = int(max_targets)
        self.scale = float(batch_size if loss_scale is None else loss_scale)     # exp.py:104-138: total = B * sum
        self.x = torch.zeros((batch_size, 3, height, width), dtype=torch.float32, device=dev)
        # boxes f64 [cap, 4] | labels i64 [cap] | sample ids i32 [cap] as views of ONE device block: a batch's targets arrive
        # as one host -> device copy of the pinned mirror block (three copies + three fills cost 0.17 ms per step)
        cap = self.cap
        tbytes = (cap * 44 + 15) // 16 * 16            # (kodhip_pull_from_host moves 16-byte pieces)
        self._tblock = torch.zeros(tbytes, dtype=torch.uint8, device=dev)
        self.boxes = self._tblock[:cap * 32].view(torch.float64).view(cap, 4)
        self.labels = self._tblock[cap * 32:cap * 40].view(torch.int64)
        self.samples = self._tblock[cap * 40:cap * 44].view(torch.int32)
        self.targets = BatchedTargets(self.boxes, self.labels, self.samples, self.cap)
        self.shape = FeatureShape(width=width, height=height)
        self.params = list(net.parameters())
        self.graph = None
        self.total = None
        self.parts = None
        self._host = []
        for _ in range(4):
            blk = torch.zeros(tbytes, dtype=torch.uint8).pin_memory()
            self._host.append((blk, blk[:cap * 32].view(torch.float64).view(cap, 4), blk[cap * 32:cap * 40].view(torch.int64),
                               blk[cap * 40:cap * 44].view(torch.int32)))
        self._events, self._slot, self._used = [None] * 4, 0, [0] * 4

    def input_buffer(self) -> torch.Tensor:
        """input_pairs=True: the network's own input buffer (bf16 pixel pairs [B, H, W/2, 8]).  A data pipeline that runs on
        the step's stream may write the next batch straight into it (DeviceTrainPipeline.compose_host_batch(pairs_out=...))
        and pass the same tensor to __call__: no copy between pipeline and step."""
        assert self.input_pairs
        return self.eng.image_buffer(self.B, self.H, self.W)

    # -- the step itself (what gets captured)
    def _step(self):
        for p in self.params:
            p.grad = None
        total, lr = self.net.train_step(self.x, self.loss, self.shape, self.targets, self.scale, image_ready=self.input_pairs)
        self.eng.wait_grads()
        self.eng.sgd_step_device()
        return total, (lr.localization.detach(), lr.objectness.detach(), lr.classification.detach())

    def _load(self, images: torch.Tensor, targets):
        if self.input_pairs:
            buf = self.eng.image_buffer(self.B, self.H, self.W)
            assert tuple(images.shape) == tuple(buf.shape) and images.dtype == buf.dtype, (images.shape, images.dtype, buf.shape)
            if images.data_ptr() != buf.data_ptr():        # (a pipeline may composite straight into the input buffer: input_buffer())
                buf.copy_(images, non_blocking=True)
        else:
            assert tuple(images.shape) == tuple(self.x.shape), (images.shape, self.x.shape)
            self.x.copy_(images, non_blocking=True)
        if isinstance(targets, BatchedTargets):
            n = targets.n
            if n > self.cap:
                raise ValueError(f"{n} target boxes in the batch exceed the graph's capacity {self.cap}")
            self.boxes.zero_(); self.labels.zero_(); self.samples.zero_()      # padding = zero-size boxes: never assigned
            if n:
                self.boxes[:n].copy_(targets.boxes)
                self.labels[:n].copy_(targets.labels)
                self.samples[:n].copy_(targets.samples)
            return
        # host-side targets (tuple of DetectionTarget on the CPU, or the flat arrays of data.device_pipeline.PackedTargets):
        # pack them into one pinned block and upload it without stalling the host behind the previous step (a pageable
        # copy would)
        packed = hasattr(targets, "samples") and hasattr(targets, "counts")
        lens = None if packed else [int(t.boxes.shape[0]) for t in targets]
        n = int(len(targets.labels)) if packed else sum(lens)
        if n > self.cap:
            raise ValueError(f"{n} target boxes in the batch exceed the graph's capacity {self.cap}")
        k = self._slot
        self._slot = (self._slot + 1) % len(self._host)
        if self._events[k] is not None:
            self._events[k].synchronize()
        blk, hb, hl, hs = self._host[k]
        m_old = self._used[k]                  # padding = zero-size boxes: only what the slot's previous batch filled needs clearing
        if m_old > n:
            hb[n:m_old].zero_(); hl[n:m_old].zero_(); hs[n:m_old].zero_()
        self._used[k] = n
        if packed:
            if n:
                hb.numpy()[:n] = targets.boxes
                hl.numpy()[:n] = targets.labels
                hs.numpy()[:n] = targets.samples
            targets = ()
        o = 0
        for i, t in enumerate(targets):
            m = lens[i]
            if m:
                hb[o:o + m] = t.boxes.reshape(-1, 4).to(torch.float64)
                hl[o:o + m] = t.labels.reshape(-1).to(torch.int64)
                hs[o:o + m] = i
                o += m
        _lib.check(self.eng.lib.kodhip_pull_from_host(self._tblock.data_ptr(), blk.data_ptr(), blk.numel(),
                                                      torch.cuda.current_stream().cuda_stream), "pull_from_host")
        ev = torch.cuda.Event()
        ev.record()
        self._events[k] = ev

    def capture(self, images: torch.Tensor, targets, warmup: int = 2, preserve_state: bool = True):
        """Runs `warmup` eager steps on this batch (allocations, lazy initialisation), then captures the step.
        preserve_state: parameters, momentum and BatchNorm buffers are restored afterwards, so that capturing
        does not move the training trajectory."""
        eng = self.eng
        eng.pin_shape(self.B, self.H, self.W)      # the graph bakes this shape's buffer addresses in (Engine.allocate)
        self._load(images, targets)
        keep = [t.clone() for t in (eng.p_arena, eng.m_arena, eng.rm_arena, eng.rv_arena, eng.nbt_arena)] if preserve_state else None
        side = torch.cuda.Stream(device=eng.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 2)):
                self._step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.total, self.parts = self._step()
        if keep is not None:
            for t, k in zip((eng.p_arena, eng.m_arena, eng.rm_arena, eng.rv_arena, eng.nbt_arena), keep):
                t.copy_(k)
            eng.mark_params_changed()
        return self

    def __call__(self, images: torch.Tensor, targets, lr: Optional[Sequence[float]] = None,
                 momentum: Optional[Sequence[float]] = None, weight_decay: Optional[Sequence[float]] = None,
                 grad_scale: float = 1.0):
        """One replayed step on a new batch.  Returns (total, (box, obj, cls)) - static tensors that the next call
        overwrites (clone to keep)."""
        if self.graph is None:
            raise RuntimeError("call capture() first")
        if lr is not None:
            self.eng.set_hyper(lr, momentum, weight_decay, grad_scale)
        if self.eng.peer is not None:
            self.eng.peer.check()             # a failed SyncBN exchange of an earlier replay: raise, do not train on NaN
        self._load(images, targets)
        self.graph.replay()
        self.eng.param_version += 1           # the replayed SGD changed the parameters
        self.eng.note_sgd_step()
        return self.total, self.parts


class GraphedEvalForward:
    """Eval-mode forward + prediction decode captured once per input shape and replayed (validation loop).

    Issued one launch at a time from Python the eval forward is host-bound like the training step (~230 launches,
    ~15 ms of ctypes / Python per batch against ~7 ms of GPU time); replaying a hipGraph leaves the host only the copy
    of the batch into the static input buffer.  Weight packs and the eval-mode BatchNorm constants are refreshed
    eagerly before a replay when parameters / running statistics moved (both live in buffers with fixed addresses), so
    the same graph serves every validation epoch."""

    def __init__(self, net, anchor_info, batch_size: int, height: int, width: int):
        from ..lightning.experiments.yv5_baseline.layers import get_detections
        self.net, self.anchor_info, self._decode = net, anchor_info, get_detections
        self.eng = net.engine()
        self.x = torch.zeros((batch_size, 3, height, width), dtype=torch.float32, device=self.eng.device)
        self.shape = FeatureShape(width=width, height=height)
        self.graph, self.det = None, None

    def _forward(self):
        from ..nn.networks.yolov5 import Yolov5NetworkResult
        from ..nn.heads.types import DetectionHeadResult
        raws = self.eng.forward(self.x, training=False)
        res = Yolov5NetworkResult(*[DetectionHeadResult(t[..., 0:4], t[..., 4:5], t[..., 5:]) for t in raws])
        return self._decode(self.shape, res, self.anchor_info)

    def _refresh(self):
        eng = self.eng
        if eng._packed_version != eng.param_version:
            eng.pack_weights()
        eng._eval_affine_ptrs()

    @torch.no_grad()
    def capture(self, images: torch.Tensor):
        eng = self.eng
        eng.pin_shape(*[self.x.shape[0], self.x.shape[2], self.x.shape[3]])
        self.x.copy_(images)
        side = torch.cuda.Stream(device=eng.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self._forward()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._refresh()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.det = self._forward()
        return self

    @torch.no_grad()
    def __call__(self, images: torch.Tensor) -> torch.Tensor:
        """Decoded detections [B, rows, 5 + nc] of a new batch (a static tensor the next call overwrites)."""
        if self.graph is None:
            self.capture(images)
        assert tuple(images.shape) == tuple(self.x.shape), (images.shape, self.x.shape)
        self._refresh()
        self.x.copy_(images, non_blocking=True)
        # a replay needs this shape's buffer set to be what the graph captured - it is pinned, nothing to swap
        self.graph.replay()
        return self.det
