"""Oracle: COCO-style mean average precision as the reference's validation callback reports it.

TEST INFRASTRUCTURE - see oracle/__init__.py.

PARITY UNPINNED.  The reference computes mAP with
``vision_evaluation.evaluators.CocoMeanAveragePrecisionEvaluator(ious=[0.3, 0.5, 0.75, 0.9],
report_tag_wise=[False, True, False, False])`` (kod/lightning/callbacks/pycoco_map_eval.py:45-48,106-125),
an un-vendored dependency (requirements.txt:29, unpinned) that wraps pycocotools' COCOeval; neither is in
the reference tree or installed here.  This file restates pycocotools 2.0's published algorithm
(cocoeval.py: evaluateImg / accumulate / summarize, bbox IoU without +1, area range "all", maxDets 100,
101 recall points) from general knowledge; ``avg_mAP`` is taken as the mean over the four requested
thresholds.
"""
from __future__ import annotations

import numpy as np

IOUS = (0.3, 0.5, 0.75, 0.9)
MAX_DETS = 100
REC_THRS = np.linspace(0.0, 1.0, 101)


def box_iou(d: np.ndarray, g: np.ndarray) -> np.ndarray:
    """[nd,4] x [ng,4] xyxy -> [nd,ng] (float64)."""
    d, g = d.astype(np.float64), g.astype(np.float64)
    ad = (d[:, 2] - d[:, 0]) * (d[:, 3] - d[:, 1])
    ag = (g[:, 2] - g[:, 0]) * (g[:, 3] - g[:, 1])
    w = np.clip(np.minimum(d[:, None, 2], g[None, :, 2]) - np.maximum(d[:, None, 0], g[None, :, 0]), 0, None)
    h = np.clip(np.minimum(d[:, None, 3], g[None, :, 3]) - np.maximum(d[:, None, 1], g[None, :, 1]), 0, None)
    inter = w * h
    return inter / (ad[:, None] + ag[None, :] - inter)


def match_image(det: np.ndarray, gt_boxes: np.ndarray, gt_labels: np.ndarray, nc: int, ious=IOUS):
    """COCOeval.evaluateImg for every category.  det: [n,6] (xyxy, score, cls) in descending score order.
    Returns list over classes of (scores [m], matched [T,m] bool, n_gt)."""
    out = []
    for c in range(nc):
        d = det[det[:, 5] == c]
        order = np.argsort(-d[:, 4], kind="mergesort")[:MAX_DETS]
        d = d[order]
        g = gt_boxes[gt_labels == c]
        matched = np.zeros((len(ious), len(d)), dtype=bool)
        if len(d) and len(g):
            iou = box_iou(d[:, :4], g)
            for ti, t in enumerate(ious):
                gtm = np.zeros(len(g), dtype=bool)
                for di in range(len(d)):
                    best, m = min(t, 1 - 1e-10), -1
                    for gi in range(len(g)):
                        if gtm[gi]:
                            continue
                        if iou[di, gi] < best:
                            continue
                        best, m = iou[di, gi], gi
                    if m >= 0:
                        gtm[m] = True
                        matched[ti, di] = True
        out.append((d[:, 4].astype(np.float64), matched, len(g)))
    return out


def accumulate(per_image, nc: int, ious=IOUS):
    """COCOeval.accumulate + summarize: AP[T, nc] (nan where a class has no ground truth)."""
    ap = np.full((len(ious), nc), np.nan)
    for c in range(nc):
        scores = np.concatenate([img[c][0] for img in per_image]) if per_image else np.zeros(0)
        npig = sum(img[c][2] for img in per_image)
        if npig == 0:
            continue
        order = np.argsort(-scores, kind="mergesort")
        for ti in range(len(ious)):
            tp = np.concatenate([img[c][1][ti] for img in per_image])[order] if len(scores) else np.zeros(0, bool)
            tps, fps = np.cumsum(tp).astype(np.float64), np.cumsum(~tp).astype(np.float64)
            rc = tps / npig
            pr = tps / (fps + tps + np.spacing(1))
            for i in range(len(pr) - 1, 0, -1):
                if pr[i] > pr[i - 1]:
                    pr[i - 1] = pr[i]
            q = np.zeros(len(REC_THRS))
            inds = np.searchsorted(rc, REC_THRS, side="left")
            for ri, pi in enumerate(inds):
                if pi < len(pr):
                    q[ri] = pr[pi]
            ap[ti, c] = q.mean()
    return ap


def report(ap: np.ndarray, class_names=None, ious=IOUS):
    """The dictionary PyCOCOMAPEvalCallback logs (pycoco_map_eval.py:113-142)."""
    per_thr = np.nanmean(ap, axis=1)
    res = {"map": float(per_thr.mean())}
    for t, v in zip(ious, per_thr):
        res[f"map{int(round(t * 100))}"] = float(v)
    i50 = list(ious).index(0.5)
    for c in range(ap.shape[1]):
        name = class_names[c] if class_names else str(c)
        res[f"map50_{name}"] = float(ap[i50, c])
    return res
