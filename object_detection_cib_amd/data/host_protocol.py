"""Host side of the training data protocol - numpy only (no torch, no GPU, no libkodhip): what a data worker process runs.

The reference's per-sample protocol (index choice, RNG draw order, box arithmetic in numpy f64) turned into ONE
compositing descriptor per sample for the HIP kernel (csrc/compose.hip).  Mirrors (same draw order, same arithmetic):
  DetectionDataset.__getitem__          kod/data/detection.py:102-156
  MosaicAugmentor.__call__              kod/data/mosaic.py:51-161
  TrainSampleAugmentor.__call__         kod/data/augmentations/default.py:440-488
  random_perspective / boxes / flip     kod/data/augmentations/default.py:111-351,386-397 (affine and perspective warps)
  mixup                                 kod/data/augmentations/default.py:400-408
data/device_pipeline.py runs HostProtocol in-process and owns the device half; data/producer.py runs it in a worker process.
"""
from __future__ import annotations

import math
import random as _random
from typing import List, NamedTuple, Sequence, Tuple

import numpy as np


class AffineParams(NamedTuple):            # default.py:31-58 / configs/data/augmentations/aug_params.yaml
    degrees: float = 0.0
    translate: float = 0.1
    scale: float = 0.5
    shear: float = 0.0
    perspective: float = 0.0


class HSVParams(NamedTuple):               # default.py:61-77
    hue: float = 0.015
    saturation: float = 0.7
    value: float = 0.4


class AugParams(NamedTuple):               # default.py:80-108
    affine_params: AffineParams = AffineParams()
    hsv_params: HSVParams = HSVParams()
    flip_lr_prob: float = 0.5
    image_color_transforms: bool = True    # (the reference's default, default.py:85 / aug_params.yaml:15)


# ----------------------------------------------------------------------------- host box / matrix math (numpy f64)
def _box_candidates(orig, proc, eps, wh_thr=2.0, ar_thr=20.0, area_thr=0.1):
    w1, h1 = orig[2] - orig[0], orig[3] - orig[1]
    w2, h2 = proc[2] - proc[0], proc[3] - proc[1]
    ar = np.maximum(w2 / (h2 + eps), h2 / (w2 + eps))
    return (w2 > wh_thr) & (h2 > wh_thr) & (w2 * h2 / (w1 * h1 + eps) > area_thr) & (ar < ar_thr)


def mosaic_layout(shapes: Sequence[Tuple[int, int]], xc: int, yc: int, S: int):
    """Destination / source rectangles of the four tiles (mosaic.py:71-130)."""
    rects = []
    for i, (h, w) in enumerate(shapes):
        if i == 0:
            a = (max(xc - w, 0), max(yc - h, 0), xc, yc)
            b = (w - (a[2] - a[0]), h - (a[3] - a[1]))
        elif i == 1:
            a = (xc, max(yc - h, 0), min(xc + w, 2 * S), yc)
            b = (0, h - (a[3] - a[1]))
        elif i == 2:
            a = (max(xc - w, 0), yc, xc, min(2 * S, yc + h))
            b = (w - (a[2] - a[0]), 0)
        else:
            a = (xc, yc, min(xc + w, 2 * S), min(2 * S, yc + h))
            b = (0, 0)
        rects.append((a, b))
    return rects


def mosaic_boxes(samples, rects, S: int):
    """Shift / filter / clip of mosaic.py:134-153 (including the stale-box quirk on box-less tiles)."""
    bbs, lbs = [], []
    shifted = None
    for (_, boxes, labels), (a, b) in zip(samples, rects):
        if len(boxes) > 0:
            shifted = boxes.copy()
            shifted[:, [0, 2]] += a[0] - b[0]
            shifted[:, [1, 3]] += a[1] - b[1]
        bbs.append(shifted)
        lbs.append(labels)
    bb, lb = np.concatenate(bbs, 0), np.concatenate(lbs, 0)
    keep = _box_candidates(bb.T, np.clip(bb, 0, 2 * S).T, 1e-7)
    return np.clip(bb[keep], 0, 2 * S - 1), lb[keep]


def affine_matrix(draws, w_in: int, h_in: int, border):
    px, py, deg, sc, shx, shy, tx, ty = draws
    w_out, h_out = w_in + 2 * border[1], h_in + 2 * border[0]
    C = np.eye(3); C[0, 2] = -w_in / 2; C[1, 2] = -h_in / 2
    P = np.eye(3); P[2, 0] = px; P[2, 1] = py
    a = math.radians(deg)
    R = np.eye(3)
    R[0, 0] = R[1, 1] = sc * math.cos(a)
    R[0, 1] = sc * math.sin(a); R[1, 0] = -sc * math.sin(a)
    Sh = np.eye(3); Sh[0, 1] = math.tan(shx * math.pi / 180); Sh[1, 0] = math.tan(shy * math.pi / 180)
    T = np.eye(3); T[0, 2] = tx * w_out; T[1, 2] = ty * h_out
    return T @ Sh @ R @ P @ C, w_out, h_out


def invert_affine(M):
    """The inverse OpenCV's warpAffine builds from the forward 2x3 matrix (imgwarp.cpp)."""
    m = np.array(M[:2], dtype=np.float64).copy()
    D = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[1, 1] * D, m[0, 0] * D
    m[0, 0] = A11; m[0, 1] *= -D; m[1, 0] *= -D; m[1, 1] = A22
    b1 = -m[0, 0] * m[0, 2] - m[0, 1] * m[1, 2]
    b2 = -m[1, 0] * m[0, 2] - m[1, 1] * m[1, 2]
    m[0, 2], m[1, 2] = b1, b2
    return m


def affine_boxes(boxes, M, w_out, h_out, scale, perspective: bool = False):
    """_process_affine_bboxes + _box_candidates (default.py:249-276,323-345); perspective: the corners' divide by w"""
    n = len(boxes)
    xy = np.ones((n * 4, 3))
    xy[:, :2] = boxes[:, [0, 1, 2, 3, 0, 3, 2, 1]].reshape(n * 4, 2)
    xy = xy @ M.T
    xy = (xy[:, :2] / xy[:, 2:3] if perspective else xy[:, :2]).reshape(n, 8)
    x, y = xy[:, [0, 2, 4, 6]], xy[:, [1, 3, 5, 7]]
    nb = np.concatenate((x.min(1), y.min(1), x.max(1), y.max(1))).reshape(4, n).T
    nb[:, [0, 2]] = nb[:, [0, 2]].clip(0, w_out - 1)
    nb[:, [1, 3]] = nb[:, [1, 3]].clip(0, h_out - 1)
    return nb, _box_candidates(boxes.T * scale, nb.T, 1e-16)


_TILE = np.dtype([("off", "<i8"), ("h", "<i4"), ("w", "<i4"), ("x1a", "<i4"), ("y1a", "<i4"), ("x2a", "<i4"),
                  ("y2a", "<i4"), ("x1b", "<i4"), ("y1b", "<i4")], align=True)
SAMPLE_DESC = np.dtype([("tile", _TILE, (4,)), ("im", "<f8", (6,)), ("pw", "<f8", (3,)), ("lut_h", "u1", (256,)), ("lut_s", "u1", (256,)),
                        ("lut_v", "u1", (256,)), ("hsv_on", "<i4"), ("flip", "<i4"), ("canvas", "<i4"), ("persp", "<i4"),
                        ("color", "<i4"), ("blur_k", "<i4"), ("median_k", "<i4"), ("pad0", "<i4"), ("pre", "<u8"), ("clahe_clip", "<f8")], align=True)

# ---- image_color_transforms (default.py:420-432,460-461): A.Compose([A.Blur(p=0.01), A.MedianBlur(p=0.01), A.ToGray(p=0.01),
# A.CLAHE(p=0.01)]) on the warped u8 image, before the HSV jitter.  The gate below is the library's draw protocol (Compose: one
# draw against its p = 1; every transform: one draw against its p; Blur / MedianBlur: a choice among the odd kernel sizes 3..7;
# CLAHE: uniform(1, 4.0) clip limit, 8 x 8 tiles) on a generator of the stage's OWN (albumentations >= 1.4 keeps one per
# Compose; 1.3.x drew from python's global `random`, interleaving with DetectionDataset's index draws - the reference does
# not pin the version, requirements.txt:26).  albumentations is not installed here: protocol and pixel arithmetic are
# restatements (parity unpinned); what tests/golden/protocol.npz pins is the stage's position.
COLOR_BLUR, COLOR_MEDIAN, COLOR_GRAY, COLOR_CLAHE = 1, 2, 4, 8


def color_gate(g: "_random.Random", p: float = 0.01):
    """(ops bit mask, blur ksize, median ksize, CLAHE clip limit) of one call of the colour Compose"""
    ops, kb, km, clip = 0, 0, 0, 0.0
    g.random()
    if g.random() < p:
        ops |= COLOR_BLUR
        kb = int(g.choice(list(range(3, 8, 2))))
    if g.random() < p:
        ops |= COLOR_MEDIAN
        km = int(g.choice(list(range(3, 8, 2))))
    if g.random() < p:
        ops |= COLOR_GRAY
    if g.random() < p:
        ops |= COLOR_CLAHE
        clip = float(g.uniform(1, 4.0))
    return ops, kb, km, clip


def augment_into(desc, aug: AugParams, rng: np.random.Generator, bb, lb, canvas: int, border, always_warp: bool = False,
                 color_rng=None, albu13: bool = False):
    """TrainSampleAugmentor.__call__ (default.py:440-488) for one sample: consumes the augmentor's generator in the
    reference's order (8 affine draws, 3 HSV draws, 1 flip draw), writes the pixel-side parameters (inverse affine
    matrix, colour-stage draws, HSV LUTs, flip flag) into the compositing descriptor and returns the transformed boxes /
    labels and the output image side.  color_rng: the colour stage's generator (needed when aug.image_color_transforms).
    albu13: model albumentations 1.3.x, whose Compose / transform gates draw on python's GLOBAL generator - the colour
    stage then draws there (color_rng is ignored) and the ToFloat / ToTensorV2 Compose at the end of the call makes three
    draws (tests/golden/protocol.npz case 'albu13')."""
    ap = aug.affine_params
    desc["persp"] = 0
    if not always_warp and ap.degrees == 0.0 and ap.translate == 0.0 and ap.scale == 0.0 and ap.shear == 0.0 and ap.perspective == 0.0:
        # AffineParams.should_aug() is False (default.py:38-48,445-457): no draws, no warp, the image keeps its size
        wo = ho = canvas
        desc["im"] = np.array([1.0, 0.0, 0.0, 0.0, 1.0, 0.0])
    else:
        draws = (rng.uniform(-ap.perspective, ap.perspective), rng.uniform(-ap.perspective, ap.perspective),
                 rng.uniform(-ap.degrees, ap.degrees), rng.uniform(1 - ap.scale, 1 + ap.scale),
                 rng.uniform(-ap.shear, ap.shear), rng.uniform(-ap.shear, ap.shear),
                 rng.uniform(0.5 - ap.translate, 0.5 + ap.translate), rng.uniform(0.5 - ap.translate, 0.5 + ap.translate))
        M, wo, ho = affine_matrix(draws, canvas, canvas, border)
        persp = draws[0] != 0 or draws[1] != 0            # default.py:306-320: cv2.warpPerspective iff a perspective draw is non-zero
        if persp:
            inv = np.linalg.inv(M)                        # (cv2.warpPerspective inverts the 3 x 3 matrix; oracle/datapath.py does the same call)
            desc["im"], desc["pw"], desc["persp"] = inv[:2].reshape(-1), inv[2], 1
        else:
            desc["im"] = invert_affine(M).reshape(-1)
        if len(lb):
            nb, keep = affine_boxes(bb, M, wo, ho, draws[3], perspective=persp)
            bb, lb = nb[keep], lb[keep]
    desc["color"], desc["blur_k"], desc["median_k"], desc["clahe_clip"], desc["pre"] = 0, 0, 0, 0.0, 0
    if aug.image_color_transforms:                       # default.py:460-461: between the warp and the HSV jitter
        assert albu13 or color_rng is not None, "image_color_transforms=True needs the colour stage's generator"
        desc["color"], desc["blur_k"], desc["median_k"], desc["clahe_clip"] = color_gate(_random if albu13 else color_rng)
    hp = aug.hsv_params
    if hp.hue == 0.0 and hp.saturation == 0.0 and hp.value == 0.0:
        desc["hsv_on"] = 0
    else:
        r = rng.uniform(-1, 1, 3) * [hp.hue, hp.saturation, hp.value] + 1               # default.py:365-369
        x = np.arange(0, 256, dtype=np.int16)
        desc["lut_h"] = ((x * r[0]) % 180).astype(np.uint8)
        desc["lut_s"] = np.clip(x * r[1], 0, 255).astype(np.uint8)
        desc["lut_v"] = np.clip(x * r[2], 0, 255).astype(np.uint8)
        desc["hsv_on"] = 1
    flip = aug.flip_lr_prob > 0.0 and rng.random() < aug.flip_lr_prob
    desc["flip"] = int(flip)
    if flip and len(bb):
        f = bb.copy()
        f[:, 2] = wo - 1 - bb[:, 0]
        f[:, 0] = wo - 1 - bb[:, 2]
        bb = f
    desc["canvas"] = canvas
    if albu13:                                            # default.py:433-438,482: tensor_transform's Compose, ToFloat, ToTensorV2 gates
        for _ in range(3):
            _random.random()
    return bb, lb, wo


class HostProtocol:
    """The HOST side of the training data protocol for one stream of batches: index choice, the reference's RNG draw
    order (`random` for mosaic partners / centres / mixup, `numpy.random` for the mixup ratio, the augmentor's own
    `default_rng(51)`), box arithmetic in numpy f64 - and nothing else: no torch, no GPU.  It turns a list of sample indices
    into the compositing descriptors + per-sample boxes / labels.  DeviceTrainPipeline runs it in-process;
    data/producer.py runs the same object in a worker process (the shape of the reference's DataLoader workers,
    kod/lightning/data_module.py:135-144) so that its 8 - 9 ms per batch of 64 leave the training step's process."""

    def __init__(self, shapes, offsets, boxes, labels, target_image_size: int, aug_params: AugParams = AugParams(),
                 mixup_prob: float = 0.0, rng_seed: int = 51, image_repeat_factors=None, sampler_indices=None,
                 albumentations_global_random: bool = False):
        self.S = int(target_image_size)
        # which albumentations generation to follow (the reference does not pin it): False = every Compose on a generator
        # of its own (>= 1.4: only the colour stage's draws exist for anyone else, on self.color_rng); True = 1.3.x, whose
        # gates draw on python's global generator and so shift the index draws that follow
        self.albu13 = bool(albumentations_global_random)
        self.shapes, self.offsets = list(shapes), np.asarray(offsets, dtype=np.int64)
        self.boxes, self.labels = list(boxes), list(labels)
        self.aug = aug_params
        self.mixup_prob = mixup_prob
        self.rng = np.random.default_rng(rng_seed)            # default.py:415
        self.color_rng = _random.Random(rng_seed) if aug_params.image_color_transforms else None     # the colour Compose's own stream
        self.weights = image_repeat_factors
        self.sampler_indices = sampler_indices if sampler_indices is not None else range(len(self.shapes))

    # -- one composite (mosaic + augment): fills a descriptor, returns boxes / labels --------------------
    def _composite(self, indices: List[int], desc):
        S = self.S
        border = (-S // 2, -S // 2)
        yc, xc = (int(_random.uniform(-x, 2 * S + x)) for x in border)          # mosaic.py:58-62
        shapes = [self.shapes[i] for i in indices]
        rects = mosaic_layout(shapes, xc, yc, S)
        samples = [(None, self.boxes[i], self.labels[i]) for i in indices]
        bb, lb = mosaic_boxes(samples, rects, S)
        for t, (i, (a, b)) in enumerate(zip(indices, rects)):
            d = desc["tile"][t]
            d["off"], d["h"], d["w"] = self.offsets[i], shapes[t][0], shapes[t][1]
            d["x1a"], d["y1a"], d["x2a"], d["y2a"] = a
            d["x1b"], d["y1b"] = b
        # always_warp: a batch is S x S, so the affine stage (which crops the 2S canvas to S) runs even when no jitter
        # is configured (the reference would hand 2S x 2S images to the collate function in that case)
        bb, lb, _ = augment_into(desc, self.aug, self.rng, bb, lb, 2 * S, border, always_warp=True, color_rng=self.color_rng,
                                 albu13=self.albu13)
        return bb, lb

    def batch(self, batch_indices: Sequence[int]):
        """(descs [B][2] SAMPLE_DESC, mix [B][2] f32, [(boxes f64 [n,4], labels i64 [n]) per sample])"""
        B = len(batch_indices)
        descs = np.zeros((B, 2), dtype=SAMPLE_DESC)
        mix = np.zeros((B, 2), dtype=np.float32)
        mix[:, 0] = -1.0
        out = []
        for k, idx in enumerate(batch_indices):
            indices = [idx] + _random.choices(self.sampler_indices, k=3, weights=self.weights)   # detection.py:119-123
            _random.shuffle(indices)
            bb, lb = self._composite(indices, descs[k, 0])
            if _random.random() < self.mixup_prob:                                               # detection.py:134-145
                m_idx = _random.choices(self.sampler_indices, k=4, weights=self.weights)
                bb2, lb2 = self._composite(m_idx, descs[k, 1])
                r = np.random.beta(32.0, 32.0)                                                   # default.py:403
                mix[k] = (np.float32(r), np.float32(1 - r))
                bb, lb = np.concatenate((bb, bb2), 0), np.concatenate((lb, lb2), 0)
            out.append((np.ascontiguousarray(bb, dtype=np.float64).reshape(-1, 4), np.ascontiguousarray(lb, dtype=np.int64).reshape(-1)))
        return descs, mix, out


class PackedTargets(NamedTuple):
    """A batch's targets as three flat host arrays (all boxes of the batch in sample order + the sample index of each):
    what engine/graphed.py copies into its pinned staging block as-is (data/producer.py hands these over from shared memory)."""
    boxes: np.ndarray        # f64 [n, 4]
    labels: np.ndarray       # i64 [n]
    samples: np.ndarray      # i32 [n]
    counts: np.ndarray       # i32 [B] boxes per sample

    def as_targets(self):
        import torch
        from .detection import DetectionTarget
        o, out = 0, []
        for c in self.counts:
            out.append(DetectionTarget(torch.from_numpy(self.boxes[o:o + c].copy()), torch.from_numpy(self.labels[o:o + c].copy())))
            o += int(c)
        return tuple(out)


def pack_targets(per_sample) -> PackedTargets:
    counts = np.array([len(lb) for _, lb in per_sample], dtype=np.int32)
    n = int(counts.sum())
    boxes = np.concatenate([bb for bb, _ in per_sample], 0).reshape(n, 4) if n else np.zeros((0, 4), np.float64)
    labels = np.concatenate([lb for _, lb in per_sample], 0) if n else np.zeros(0, np.int64)
    return PackedTargets(boxes, labels, np.repeat(np.arange(len(per_sample), dtype=np.int32), counts), counts)


