"""On-device mAP evaluation - the arithmetic of PyCOCOMAPEvalCallback
(kod/lightning/callbacks/pycoco_map_eval.py:41-144) without Lightning: detections never leave the GPU until
they are TP/FP flags.

    ev = DeviceMAPEvaluator(num_classes, class_names)
    ev.add_batch(targets, detections)      # = on_validation_batch_end (targets: Sequence[DetectionTarget])
    report = ev.get_report()               # = on_validation_epoch_end: map, map30/50/75/90, map50_<class>
    ev.reset()

Matching runs in csrc/map_match.hip; the precision/recall accumulation (101 recall points, pycocotools
semantics, IoUs .3/.5/.75/.9, maxDets 100) is host numpy on a few thousand rows.  Parity with
vision_evaluation/pycocotools is unpinned (not in the reference tree, not installed): DESIGN.md section 5.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np
import torch

from ... import _host, _lib
from ...core.label_assignment.yv5 import BatchedTargets

IOUS = (0.3, 0.5, 0.75, 0.9)            # pycoco_map_eval.py:45-48
MAX_DETS = 100
MAX_GT_PER_IMAGE = 256          # size of the per-lane 'matched' bitmap in map_match_kernel
REC_THRS = np.linspace(0.0, 1.0, 101)


class DeviceMAPEvaluator:
    def __init__(self, num_classes: int, class_names: Sequence[str] | None = None):
        self.nc = num_classes
        self.names = list(class_names) if class_names else [str(i) for i in range(num_classes)]
        self.reset()

    def reset(self):
        self._scores = [[] for _ in range(self.nc)]
        self._tp = [[] for _ in range(self.nc)]
        self._npig = np.zeros(self.nc, dtype=np.int64)

    def add_batch(self, targets, detections: Sequence[torch.Tensor]):
        """detections: list of [n<=300, 6] tensors (non_max_suppression output, descending score)."""
        _lib.require_gpu()
        dev = detections[0].device if len(detections) else torch.device("cuda")
        B = len(detections)
        max_det = max([int(d.shape[0]) for d in detections] + [1])
        det = torch.zeros((B, max_det, 6), dtype=torch.float32, device=dev)
        ndet = torch.tensor([int(d.shape[0]) for d in detections], dtype=torch.int32, device=dev)
        for b, d in enumerate(detections):
            if d.shape[0]:
                det[b, :d.shape[0]] = d
        bt = targets if isinstance(targets, BatchedTargets) else BatchedTargets.from_targets(targets, dev)
        counts = torch.bincount(bt.samples.long(), minlength=B) if bt.n else torch.zeros(B, dtype=torch.long, device=dev)
        most = int(_host.fetch(counts.max())[0]) if bt.n else 0
        if most > MAX_GT_PER_IMAGE:
            raise ValueError(f"an image carries {most} ground-truth boxes; the device matcher tracks at most "
                             f"{MAX_GT_PER_IMAGE} per image (csrc/map_match.hip)")
        start = torch.zeros(B + 1, dtype=torch.int32, device=dev)
        start[1:] = torch.cumsum(counts, 0).int()
        T = len(IOUS)
        tp = torch.zeros((B, max_det, T), dtype=torch.uint8, device=dev)
        counted = torch.zeros((B, max_det), dtype=torch.uint8, device=dev)
        thr = (C.c_double * T)(*IOUS)
        _lib.check(_lib.lib().kodhip_map_match(det.data_ptr(), ndet.data_ptr(), bt.boxes.data_ptr() if bt.n else None,
                                               bt.labels.data_ptr() if bt.n else None, start.data_ptr(),
                                               tp.data_ptr(), counted.data_ptr(), B, max_det, self.nc, thr, T,
                                               MAX_DETS, torch.cuda.current_stream().cuda_stream), "map_match")
        labels_dev = bt.labels if bt.n else torch.zeros(1, dtype=torch.int64, device=dev)
        det_t, tp_t, cnt_t, nd_t, lab_t = _host.fetch(det, tp, counted, ndet, labels_dev)      # one polling hand-off
        det_h, tp_h, cnt_h = det_t.numpy(), tp_t.numpy().astype(bool), cnt_t.numpy().astype(bool)
        nd_h = nd_t.numpy()
        if bt.n:
            self._npig += np.bincount(lab_t.numpy(), minlength=self.nc)[:self.nc]
        for b in range(B):
            rows = np.arange(nd_h[b])
            rows = rows[cnt_h[b, :nd_h[b]]]
            cls = det_h[b, rows, 5].astype(np.int64)
            for c in np.unique(cls):
                r = rows[cls == c]
                self._scores[c].append(det_h[b, r, 4].astype(np.float64))
                self._tp[c].append(tp_h[b, r].T)                  # [T, m]

    def average_precision(self) -> np.ndarray:
        ap = np.full((len(IOUS), self.nc), np.nan)
        for c in range(self.nc):
            if self._npig[c] == 0:
                continue
            scores = np.concatenate(self._scores[c]) if self._scores[c] else np.zeros(0)
            tps_all = np.concatenate(self._tp[c], axis=1) if self._tp[c] else np.zeros((len(IOUS), 0), bool)
            order = np.argsort(-scores, kind="mergesort")
            for ti in range(len(IOUS)):
                tp = tps_all[ti][order]
                tps, fps = np.cumsum(tp).astype(np.float64), np.cumsum(~tp).astype(np.float64)
                rc = tps / self._npig[c]
                pr = tps / (fps + tps + np.spacing(1))
                if len(pr):
                    pr = np.maximum.accumulate(pr[::-1])[::-1]
                inds = np.searchsorted(rc, REC_THRS, side="left")
                q = np.where(inds < len(pr), pr[np.minimum(inds, max(len(pr) - 1, 0))] if len(pr) else 0.0, 0.0)
                ap[ti, c] = float(np.mean(q))
        return ap

    def _report_of(self, ap: np.ndarray) -> dict:
        per_thr = np.nanmean(ap, axis=1)
        res = {"map": float(per_thr.mean())}
        for t, v in zip(IOUS, per_thr):
            res[f"map{int(round(t * 100))}"] = float(v)
        for c in range(self.nc):
            res[f"map50_{self.names[c]}"] = float(ap[IOUS.index(0.5), c])
        return res

    def gather(self, process_group=None):
        """Merge the match records of every rank into this evaluator (rank order => identical state everywhere):
        the exact whole-validation-set mAP under data-parallel validation."""
        import torch.distributed as dist
        world = dist.get_world_size(process_group)
        if world == 1:
            return self
        mine = ([[np.asarray(x) for x in c] for c in self._scores], [[np.asarray(x) for x in c] for c in self._tp], self._npig)
        parts = [None] * world
        dist.all_gather_object(parts, mine, group=process_group)
        self.reset()
        for scores, tps, npig in parts:
            for c in range(self.nc):
                self._scores[c].extend(scores[c])
                self._tp[c].extend(tps[c])
            self._npig += npig
        return self

    def get_report(self, process_group=None, sync: str = "mean") -> dict:
        """process_group None (default): the report of THIS evaluator, no collective - also inside an initialised
        torch.distributed job, so validation on one rank only can never hang on its peers.  With an explicit process
        group (validation sharded over its ranks; `torch.distributed.group.WORLD` for the default group):
        sync="mean"   - every rank evaluates its own shard and the logged values are averaged over ranks, which is what
                        the reference does (`pl_module.log_dict(results, sync_dist=True)`, pycoco_map_eval.py:139-142);
        sync="global" - match records are gathered first (gather()), giving the mAP of the whole validation set."""
        if process_group is None:
            return self._report_of(self.average_precision())
        import torch.distributed as dist
        if sync == "global":
            return self.gather(process_group)._report_of(self.average_precision())
        rep = self._report_of(self.average_precision())
        keys = sorted(rep)
        vals = torch.tensor([rep[k] for k in keys], dtype=torch.float64)
        if dist.get_backend(process_group) == "nccl":
            vals = vals.cuda()
        dist.all_reduce(vals, group=process_group)
        vals = (vals / dist.get_world_size(process_group)).cpu()
        return {k: float(v) for k, v in zip(keys, vals)}
