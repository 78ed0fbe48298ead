"""Oracle: mosaic / affine / HSV / flip / mixup compositing (numpy, CPU).

TEST INFRASTRUCTURE - see oracle/__init__.py.  Restates:

* mosaic paste + box filter       kod/data/mosaic.py:11-161
* affine matrix, box transform    kod/data/augmentations/default.py:111-351
* flip                            kod/data/augmentations/default.py:386-397
* mixup                           kod/data/augmentations/default.py:400-408
* per-sample protocol             kod/data/detection.py:102-156,
                                  kod/data/augmentations/default.py:440-488

Pinned by golden vectors: mosaic (image + boxes), affine matrices, box
transform / candidate filter, flip, mixup.

PARITY UNPINNED (third-party arithmetic absent from the reference tree and not
installed in this image): ``cv2.warpAffine`` (INTER_LINEAR, BORDER_CONSTANT)
and ``cv2.cvtColor`` BGR<->HSV.  ``warp_affine_u8`` / ``bgr2hsv_u8`` /
``hsv2bgr_u8`` below restate OpenCV 4.x's published fixed-point algorithms
(imgwarp.cpp: INTER_BITS=5, INTER_REMAP_COEF_BITS=15, AB_BITS=10;
color_hsv.cpp: hsv_shift=12 integer tables for u8).
"""
from __future__ import annotations

import math
import random

import numpy as np


# ----------------------------------------------------------------------------- mosaic
def _candidates(orig, proc, eps, wh_thr=2.0, ar_thr=20.0, area_thr=0.1):
    """box_candidates (mosaic.py:11-35, eps 1e-7) / _box_candidates (default.py:195-216, eps 1e-16);
    both take [4,n] arrays."""
    w1, h1 = orig[2] - orig[0], orig[3] - orig[1]
    w2, h2 = proc[2] - proc[0], proc[3] - proc[1]
    ar = np.maximum(w2 / (h2 + eps), h2 / (w2 + eps))
    return (w2 > wh_thr) & (h2 > wh_thr) & (w2 * h2 / (w1 * h1 + eps) > area_thr) & (ar < ar_thr)


def mosaic_centre(S: int, rnd=random):
    """mosaic.py:58-62: yc then xc, each int(uniform(S/2, 3S/2)) (border = -S//2)."""
    border = (-S // 2, -S // 2)
    yc, xc = (int(rnd.uniform(-x, 2 * S + x)) for x in border)
    return yc, xc, border


def mosaic_rects(i: int, w: int, h: int, xc: int, yc: int, S: int):
    """Destination (a) / source (b) rectangles of tile i (mosaic.py:71-130)."""
    if i == 0:
        a = (max(xc - w, 0), max(yc - h, 0), xc, yc)
        b = (w - (a[2] - a[0]), h - (a[3] - a[1]), w, h)
    elif i == 1:
        a = (xc, max(yc - h, 0), min(xc + w, 2 * S), yc)
        b = (0, h - (a[3] - a[1]), min(w, a[2] - a[0]), h)
    elif i == 2:
        a = (max(xc - w, 0), yc, xc, min(2 * S, yc + h))
        b = (w - (a[2] - a[0]), 0, w, min(a[3] - a[1], h))
    else:
        a = (xc, yc, min(xc + w, 2 * S), min(2 * S, yc + h))
        b = (0, 0, min(w, a[2] - a[0]), min(a[3] - a[1], h))
    return a, b


def mosaic(samples, S: int, rnd=random):
    """MosaicAugmentor.__call__ (mosaic.py:51-161); samples = 4 x (u8 HWC, boxes f64, labels)."""
    yc, xc, border = mosaic_centre(S, rnd)
    canvas = np.full((2 * S, 2 * S, 3), 114, dtype=np.uint8)
    bbs, lbs = [], []
    for i, (img, boxes, labels) in enumerate(samples):
        h, w = img.shape[:2]
        a, b = mosaic_rects(i, w, h, xc, yc, S)
        canvas[a[1]:a[3], a[0]:a[2]] = img[b[1]:b[3], b[0]:b[2]]
        if len(boxes) > 0:
            shifted = boxes.copy()
            shifted[:, [0, 2]] += a[0] - b[0]
            shifted[:, [1, 3]] += a[1] - b[1]
        bbs.append(shifted)          # reference quirk: stale boxes re-appended for box-less tiles
        lbs.append(labels)
    bb = np.concatenate(bbs, 0)
    lb = np.concatenate(lbs, 0)
    keep = _candidates(bb.T, np.clip(bb, 0, 2 * S).T, eps=1e-7)
    bb = np.clip(bb[keep], 0, 2 * S - 1)
    return canvas, bb, lb[keep], border, (yc, xc)


# ----------------------------------------------------------------------------- affine
def affine_draws(rng: np.random.Generator, degrees=0.0, translate=0.1, scale=0.5, shear=0.0, perspective=0.0):
    """get_affine_random_values (default.py:111-140): 8 uniforms in the reference's order."""
    px = rng.uniform(-perspective, perspective)
    py = rng.uniform(-perspective, perspective)
    deg = rng.uniform(-degrees, degrees)
    sc = rng.uniform(1 - scale, 1 + scale)
    shx = rng.uniform(-shear, shear)
    shy = rng.uniform(-shear, shear)
    tx = rng.uniform(0.5 - translate, 0.5 + translate)
    ty = rng.uniform(0.5 - translate, 0.5 + translate)
    return px, py, deg, sc, shx, shy, tx, ty


def affine_matrix(draws, w_in: int, h_in: int, border=(0, 0)):
    """M = T.S.R.P.C (default.py:143-277); R as cv2.getRotationMatrix2D(angle, (0,0), scale)."""
    px, py, deg, sc, shx, shy, tx, ty = draws
    w_out, h_out = w_in + 2 * border[1], h_in + 2 * border[0]
    C = np.eye(3); C[0, 2] = -w_in / 2; C[1, 2] = -h_in / 2
    P = np.eye(3); P[2, 0] = px; P[2, 1] = py
    a = math.radians(deg)
    Rm = np.eye(3)
    Rm[0, 0] = Rm[1, 1] = sc * math.cos(a)
    Rm[0, 1] = sc * math.sin(a); Rm[1, 0] = -sc * math.sin(a)
    Sh = np.eye(3); Sh[0, 1] = math.tan(shx * math.pi / 180); Sh[1, 0] = math.tan(shy * math.pi / 180)
    T = np.eye(3); T[0, 2] = tx * w_out; T[1, 2] = ty * h_out
    return T @ Sh @ Rm @ P @ C, (w_out, h_out)


def affine_boxes(boxes: np.ndarray, M: np.ndarray, w_out: int, h_out: int, scale: float):
    """_process_affine_bboxes + _box_candidates (default.py:249-276,323-345), affine case."""
    n = len(boxes)
    xy = np.ones((n * 4, 3))
    xy[:, :2] = boxes[:, [0, 1, 2, 3, 0, 3, 2, 1]].reshape(n * 4, 2)
    xy = (xy @ M.T)[:, :2].reshape(n, 8)
    x, y = xy[:, [0, 2, 4, 6]], xy[:, [1, 3, 5, 7]]
    nb = np.concatenate((x.min(1), y.min(1), x.max(1), y.max(1))).reshape(4, n).T
    nb[:, [0, 2]] = nb[:, [0, 2]].clip(0, w_out - 1)
    nb[:, [1, 3]] = nb[:, [1, 3]].clip(0, h_out - 1)
    keep = _candidates(boxes.T * scale, nb.T, eps=1e-16)
    return nb, keep


def perspective_boxes(boxes: np.ndarray, M: np.ndarray, w_out: int, h_out: int, scale: float):
    """_process_affine_bboxes(perspective=True) + _box_candidates (default.py:249-276,323-345): corners through the full
    3 x 3 matrix with the perspective divide."""
    n = len(boxes)
    xy = np.ones((n * 4, 3))
    xy[:, :2] = boxes[:, [0, 1, 2, 3, 0, 3, 2, 1]].reshape(n * 4, 2)
    xy = xy @ M.T
    xy = (xy[:, :2] / xy[:, 2:3]).reshape(n, 8)
    x, y = xy[:, [0, 2, 4, 6]], xy[:, [1, 3, 5, 7]]
    nb = np.concatenate((x.min(1), y.min(1), x.max(1), y.max(1))).reshape(4, n).T
    nb[:, [0, 2]] = nb[:, [0, 2]].clip(0, w_out - 1)
    nb[:, [1, 3]] = nb[:, [1, 3]].clip(0, h_out - 1)
    keep = _candidates(boxes.T * scale, nb.T, eps=1e-16)
    return nb, keep


def flip_boxes(boxes: np.ndarray, width: int):
    """horizontal_flip (default.py:386-397)."""
    out = boxes.copy()
    if len(out):
        out[:, 2] = width - 1 - boxes[:, 0]
        out[:, 0] = width - 1 - boxes[:, 2]
    return out


def mixup_blend(im1, im2, r: float):
    """mixup (default.py:400-408); images are f32 CHW, r is a python/numpy f64 scalar."""
    return im1 * r + im2 * (1 - r)


# ----------------------------------------------------------------------------- OpenCV restatements (unpinned)
def warp_affine_u8(src: np.ndarray, M23: np.ndarray, w_out: int, h_out: int, border_value: int = 114):
    """cv2.warpAffine(src, M, (w_out,h_out), INTER_LINEAR, BORDER_CONSTANT) for u8 HWC.

    OpenCV imgwarp.cpp: invert M; per destination pixel X0 = saturate(round((M00*x)*1024)),
    fixed point AB_BITS=10; coordinates rounded to 1/32 px (INTER_BITS=5); bilinear weights from
    the 32x32 table in 15-bit integers (sum 32768); result = (sum + 16384) >> 15.
    """
    M = np.asarray(M23, dtype=np.float64)
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[1, 1] * D, M[0, 0] * D
    iM = np.array([[A11, -M[0, 1] * D, 0.0], [-M[1, 0] * D, A22, 0.0]])
    iM[0, 2] = -iM[0, 0] * M[0, 2] - iM[0, 1] * M[1, 2]
    iM[1, 2] = -iM[1, 0] * M[0, 2] - iM[1, 1] * M[1, 2]
    AB_BITS, INTER_BITS = 10, 5
    AB_SCALE = 1 << AB_BITS
    INTER_TAB = 1 << INTER_BITS
    rnd = AB_SCALE // INTER_TAB // 2
    xs = np.arange(w_out)
    adelta = np.rint(iM[0, 0] * xs * AB_SCALE).astype(np.int64)        # cvRound = rint (half-even)
    bdelta = np.rint(iM[1, 0] * xs * AB_SCALE).astype(np.int64)
    ys = np.arange(h_out)
    X0 = np.rint((iM[0, 1] * ys + iM[0, 2]) * AB_SCALE).astype(np.int64) + rnd
    Y0 = np.rint((iM[1, 1] * ys + iM[1, 2]) * AB_SCALE).astype(np.int64) + rnd
    X = (X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    sx, sy = X >> INTER_BITS, Y >> INTER_BITS
    fx, fy = X & (INTER_TAB - 1), Y & (INTER_TAB - 1)
    tab = _bilinear_tab()
    wts = tab[fy, fx]                                                  # [h,w,4] int (w00,w01,w10,w11)
    h, w = src.shape[:2]
    out = np.empty((h_out, w_out, src.shape[2]), dtype=np.uint8)

    def fetch(yy, xx):
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        v = src[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)].astype(np.int64)
        v[~ok] = border_value
        return v

    acc = (fetch(sy, sx) * wts[..., 0:1] + fetch(sy, sx + 1) * wts[..., 1:2]
           + fetch(sy + 1, sx) * wts[..., 2:3] + fetch(sy + 1, sx + 1) * wts[..., 3:4])
    out[:] = ((acc + (1 << 14)) >> 15).astype(np.uint8)
    return out


def warp_perspective_u8(src: np.ndarray, M33: np.ndarray, w_out: int, h_out: int, border_value: int = 114):
    """cv2.warpPerspective(src, M, (w_out, h_out), borderValue=114) (default flags: INTER_LINEAR, BORDER_CONSTANT) for u8 HWC.

    OpenCV imgwarp.cpp (WarpPerspectiveInvoker): M is inverted (here: numpy.linalg.inv, the same call on the product's host
    side); per destination pixel, in doubles, X0 = M00 x + M01 y + M02, Y0 likewise, W = M20 x + M21 y + M22;
    W = W ? 32 / W : 0; X = saturate_cast<int>(clamp(X0 W)), Y likewise (cvRound: half to even); source pixel
    (X >> 5, Y >> 5) saturated to int16, bilinear weights from the same 32 x 32 fixed-point table as warpAffine.
    PARITY UNPINNED like warp_affine_u8 (and OpenCV evaluates the three sums block-wise: the last bit of X0 W may differ)."""
    iM = np.linalg.inv(np.asarray(M33, dtype=np.float64))
    xs = np.arange(w_out, dtype=np.float64)[None, :]
    ys = np.arange(h_out, dtype=np.float64)[:, None]
    X0 = (iM[0, 0] * xs + iM[0, 1] * ys) + iM[0, 2]
    Y0 = (iM[1, 0] * xs + iM[1, 1] * ys) + iM[1, 2]
    W = (iM[2, 0] * xs + iM[2, 1] * ys) + iM[2, 2]
    with np.errstate(divide="ignore"):
        W = np.where(W != 0, 32.0 / W, 0.0)
    lim = lambda v: np.maximum(-2147483648.0, np.minimum(2147483647.0, v))
    X = np.rint(lim(X0 * W)).astype(np.int64)
    Y = np.rint(lim(Y0 * W)).astype(np.int64)
    sx = np.clip(X >> 5, -32768, 32767)
    sy = np.clip(Y >> 5, -32768, 32767)
    fx, fy = X & 31, Y & 31
    wts = _bilinear_tab()[fy, fx]
    h, w = src.shape[:2]

    def fetch(yy, xx):
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        v = src[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)].astype(np.int64)
        v[~ok] = border_value
        return v

    acc = (fetch(sy, sx) * wts[..., 0:1] + fetch(sy, sx + 1) * wts[..., 1:2]
           + fetch(sy + 1, sx) * wts[..., 2:3] + fetch(sy + 1, sx + 1) * wts[..., 3:4])
    return ((acc + (1 << 14)) >> 15).astype(np.uint8)


_TAB = None


def _bilinear_tab():
    """OpenCV initInterTab2D(INTER_LINEAR, fixpt=true): 32x32x(2x2) short weights = saturate(wy*wx*32768).
    Products are exact multiples of 32 => sums are 32768, except entry (0,0) whose 32768 saturates to 32767;
    OpenCV's fix-up for that entry indexes past the 2x2 block (into the next entry, later overwritten)."""
    global _TAB
    if _TAB is None:
        n = 32
        t1 = np.stack((1.0 - np.arange(n) / n, np.arange(n) / n), 1).astype(np.float32)   # [32,2]
        tab = np.zeros((n, n, 4), dtype=np.int64)
        for i in range(n):
            for j in range(n):
                f = (t1[i][:, None] * t1[j][None, :]).astype(np.float32)                  # [ky,kx]
                tab[i, j] = np.array([_sat16(np.rint(float(v) * 32768)) for v in f.reshape(-1)], dtype=np.int64)
        _TAB = tab
    return _TAB


def _sat16(v):
    return int(max(-32768, min(32767, v)))


def bgr2hsv_u8(img: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(img, COLOR_BGR2HSV) for u8 (H in [0,180)), OpenCV integer tables (hsv_shift=12)."""
    b = img[..., 0].astype(np.int64); g = img[..., 1].astype(np.int64); r = img[..., 2].astype(np.int64)
    v = np.maximum(np.maximum(b, g), r)
    vmin = np.minimum(np.minimum(b, g), r)
    diff = v - vmin
    hsv_shift = 12
    idx = np.arange(256)
    sdiv = np.zeros(256, dtype=np.int64); hdiv = np.zeros(256, dtype=np.int64)
    sdiv[1:] = np.rint((255 << hsv_shift) / idx[1:].astype(np.float64)).astype(np.int64)
    hdiv[1:] = np.rint((180 << hsv_shift) / (6.0 * idx[1:])).astype(np.int64)
    vr = (v == r); vg = (v == g)
    s = (diff * sdiv[v] + (1 << (hsv_shift - 1))) >> hsv_shift
    h = np.where(vr, g - b, np.where(vg, (b - r) + 2 * diff, (r - g) + 4 * diff))
    h = (h * hdiv[diff] + (1 << (hsv_shift - 1))) >> hsv_shift
    h = h + np.where(h < 0, 180, 0)
    return np.stack((h, s, v), -1).astype(np.uint8)


def hsv2bgr_u8(hsv: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(hsv, COLOR_HSV2BGR) for u8: OpenCV converts through float32 (h*(6/180), s/255, v/255)."""
    h = hsv[..., 0].astype(np.float32); s = hsv[..., 1].astype(np.float32) * np.float32(1 / 255.0)
    v = hsv[..., 2].astype(np.float32) * np.float32(1 / 255.0)
    hscale = np.float32(6.0 / 180.0)
    hh = h * hscale
    hh = np.where(hh < 0, hh + 6, hh); hh = np.where(hh >= 6, hh - 6, hh)
    sector = np.floor(hh).astype(np.int64)
    f = hh - sector.astype(np.float32)
    sector = np.clip(sector, 0, 5)
    t0 = v
    t1 = v * (1 - s)
    t2 = v * (1 - s * f)
    t3 = v * (1 - s * (1 - f))
    tabs = np.stack((t0, t1, t2, t3), -1)
    sector_data = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])
    sel = sector_data[sector]                                           # [...,3] -> b,g,r picks
    bgr = np.take_along_axis(tabs, sel, -1)
    bgr = np.where(s[..., None] == 0, v[..., None], bgr)
    return np.clip(np.rint(bgr * 255.0), 0, 255).astype(np.uint8)


def hsv_luts(r3: np.ndarray):
    """augment_hsv LUTs (default.py:376-379); r3 = draws*[h,s,v]+1."""
    x = np.arange(0, 256, dtype=np.int16)
    return (((x * r3[0]) % 180).astype(np.uint8), np.clip(x * r3[1], 0, 255).astype(np.uint8),
            np.clip(x * r3[2], 0, 255).astype(np.uint8))


def augment_hsv_u8(img: np.ndarray, r3: np.ndarray) -> np.ndarray:
    """augment_hsv (default.py:354-383) on an RGB image treated as BGR (reference quirk)."""
    hsv = bgr2hsv_u8(img)
    lh, ls, lv = hsv_luts(r3)
    out = np.stack((lh[hsv[..., 0]], ls[hsv[..., 1]], lv[hsv[..., 2]]), -1)
    return hsv2bgr_u8(out)


# ----------------------------------------------------------------------------- whole-sample protocol
def augment_sample(canvas, boxes, labels, border, S, rng: np.random.Generator, hsv=(0.015, 0.7, 0.4),
                   flip_prob=0.5, translate=0.1, scale=0.5, degrees=0.0, shear=0.0, perspective=0.0, log=None,
                   color=None, albu13=None):
    """TrainSampleAugmentor.__call__ (default.py:440-488).  Draw order of the augmentor's generator: 8 affine uniforms,
    3 HSV uniforms (one call), 1 flip draw - the flip draw only when flip_prob > 0 (AugParams.should_flip short-circuits,
    default.py:98-99).  color: None, or the generator (`random.Random`) of the image_color_transforms stage
    (default.py:420-432,460-461; the reference's default, configs/data/augmentations/aug_params.yaml:15) - the
    albumentations Compose between warp and HSV draws from the LIBRARY's generator, see color_gate.  log: dict that
    receives M / LUTs / flip / colour draws.  albu13: None, or python's global generator to model albumentations 1.3.x,
    whose Compose / transform gates draw on it: the colour stage then uses it too (pass it as `color`), and the ToFloat /
    ToTensorV2 Compose at the end of the call makes three draws (Compose, ToFloat, ToTensorV2)."""
    draws = affine_draws(rng, degrees=degrees, translate=translate, scale=scale, shear=shear, perspective=perspective)
    M, (wo, ho) = affine_matrix(draws, canvas.shape[1], canvas.shape[0], border)
    persp = draws[0] != 0 or draws[1] != 0                      # default.py:306-320: warpPerspective iff a perspective draw is non-zero
    img = warp_perspective_u8(canvas, M, wo, ho) if persp else warp_affine_u8(canvas, M[:2], wo, ho)
    if len(labels):
        nb, keep = (perspective_boxes if persp else affine_boxes)(boxes, M, wo, ho, draws[3])
        boxes, labels = nb[keep], labels[keep]
    cdraw = None
    if color is not None:
        cdraw = color_gate(color)
        img = apply_color_ops(img, *cdraw)
    luts = None
    if not (hsv[0] == 0.0 and hsv[1] == 0.0 and hsv[2] == 0.0):
        r3 = rng.uniform(-1, 1, 3) * list(hsv) + 1
        luts = hsv_luts(r3)
        img = augment_hsv_u8(img, r3)
    flip = bool(flip_prob > 0.0 and rng.random() < flip_prob)
    if flip:
        img = np.fliplr(img)
        boxes = flip_boxes(boxes, img.shape[1])
    if albu13 is not None:
        for _ in range(3):                                      # default.py:433-438,482: tensor_transform(image=...) under 1.3.x
            albu13.random()
    if log is not None:
        log.update(M=M, dsize=(wo, ho), luts=luts, flip=flip, color=cdraw)
    chw = np.ascontiguousarray(img.transpose(2, 0, 1)).astype(np.float32) / np.float32(255.0)
    return chw, boxes, labels


def train_sample(cache, idx, S, rng: np.random.Generator, mixup_prob=0.0, rnd=random, nprnd=np.random, weights=None,
                 sampler_indices=None, aug=None, log=None):
    """DetectionDataset.__getitem__ with mosaic on (detection.py:102-156).  cache: list of (u8 HWC, boxes, labels);
    weights / sampler_indices: the sampler's image_repeat_factors / sampler_indices side channel (detection.py:78-80,
    114-122); aug: keyword overrides of augment_sample (degrees, shear, hsv, flip_prob, ...); log: dict that receives
    the protocol's intermediate values (what tests/golden/protocol.npz records from the reference)."""
    pool = range(len(cache)) if sampler_indices is None else sampler_indices
    aug = aug or {}
    indices = [idx] + rnd.choices(pool, k=3, weights=weights)
    rnd.shuffle(indices)
    canvas, bb, lb, border, _ = mosaic([cache[i] for i in indices], S, rnd)
    l1 = {}
    mbb = bb
    img, bb, lb = augment_sample(canvas, bb, lb, border, S, rng, log=l1, **aug)
    if log is not None:
        log.update(indices=list(indices), mosaic_boxes=[mbb], stages=[l1], mixup_r=None)
    if rnd.random() < mixup_prob:
        m_idx = rnd.choices(pool, k=4, weights=weights)
        canvas2, bb2, lb2, border2, _ = mosaic([cache[i] for i in m_idx], S, rnd)
        l2 = {}
        mbb2 = bb2
        img2, bb2, lb2 = augment_sample(canvas2, bb2, lb2, border2, S, rng, log=l2, **aug)
        r = nprnd.beta(32.0, 32.0)
        import torch
        img = mixup_blend(torch.from_numpy(img), torch.from_numpy(img2), r).numpy()
        bb, lb = np.concatenate((bb, bb2), 0), np.concatenate((lb, lb2), 0)
        if log is not None:
            log["indices"] += list(m_idx)
            log["mosaic_boxes"].append(mbb2)
            log["stages"].append(l2)
            log["mixup_r"] = float(r)
    return img, bb, lb


# ----------------------------------------------------------------------------- validation pre-processing
# kod/data/sample_reader.py:16-40,102-136 (albumentations LongestMaxSize(INTER_LINEAR) + PadIfNeeded(114)) followed
# by ValidationSampleAugmentor (kod/data/augmentations/albu.py:91-119: ToFloat(255) + HWC->CHW).
# Third-party arithmetic absent from this image (albumentations 1.3.x, opencv-python 4.x): restated from their
# published algorithms, PARITY UNPINNED like the other OpenCV restatements above.
def _py3round(v: float) -> int:
    """albumentations.augmentations.geometric.functional.py3round: round half away from zero on exact .5."""
    if abs(round(v) - v) == 0.5:
        return int(2.0 * round(v / 2.0))
    return int(round(v))


def _resize_coeffs(n_src: int, n_dst: int):
    """OpenCV resize.cpp (INTER_LINEAR, 8-bit): per destination index the left source index and the two
    11-bit fixed-point weights.  fx = (float)((d + 0.5) * scale - 0.5), s = floor(fx), fx -= s; s < 0 -> (0, fx = 0);
    s >= n_src - 1 -> (n_src - 1, fx = 0) [HResize uses S[s]*2048 there, VResize clamps both rows]."""
    scale = 1.0 / (n_dst / float(n_src))                  # scale_x = 1. / inv_scale_x, inv_scale_x = dsize / ssize
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    return s, f


def resize_linear_u8(src: np.ndarray, w_out: int, h_out: int) -> np.ndarray:
    """cv2.resize(src, (w_out, h_out), interpolation=cv2.INTER_LINEAR) for u8 HWC."""
    h, w = src.shape[:2]
    sx, fx = _resize_coeffs(w, w_out)
    lo = sx < 0
    hi = sx >= w - 1
    fx = np.where(lo | hi, np.float32(0), fx)
    sx = np.where(lo, 0, np.where(hi, w - 1, sx))
    a0 = np.clip(np.rint((np.float32(1) - fx).astype(np.float64) * 2048), -32768, 32767).astype(np.int64)
    a1 = np.clip(np.rint(fx.astype(np.float64) * 2048), -32768, 32767).astype(np.int64)
    sx1 = np.minimum(sx + 1, w - 1)
    sy, fy = _resize_coeffs(h, h_out)
    b0 = np.clip(np.rint((np.float32(1) - fy).astype(np.float64) * 2048), -32768, 32767).astype(np.int64)
    b1 = np.clip(np.rint(fy.astype(np.float64) * 2048), -32768, 32767).astype(np.int64)
    r0 = np.clip(sy, 0, h - 1)
    r1 = np.clip(sy + 1, 0, h - 1)
    s = src.astype(np.int64)
    hrow = s[:, sx, :] * a0[None, :, None] + s[:, sx1, :] * a1[None, :, None]          # [h, w_out, c], 11-bit
    S0, S1 = hrow[r0], hrow[r1]
    out = ((((b0[:, None, None] * (S0 >> 4)) >> 16) + ((b1[:, None, None] * (S1 >> 4)) >> 16) + 2) >> 2)
    return np.clip(out, 0, 255).astype(np.uint8)


def val_geometry(h: int, w: int, S: int):
    """(new_h, new_w, pad_top, pad_left) of LongestMaxSize(S) + PadIfNeeded(S, S) (centre position)."""
    scale = S / float(max(w, h))
    if scale != 1.0:
        nh, nw = _py3round(h * scale), _py3round(w * scale)
    else:
        nh, nw = h, w
    top = int((S - nh) / 2.0) if nh < S else 0
    left = int((S - nw) / 2.0) if nw < S else 0
    return nh, nw, top, left


def val_sample(img: np.ndarray, boxes: np.ndarray, S: int):
    """SampleReader.__call__(letter_box=True) + ValidationSampleAugmentor: (f32 [3,S,S], boxes xyxy px f64)."""
    h, w = img.shape[:2]
    nh, nw, top, left = val_geometry(h, w, S)
    res = resize_linear_u8(img, nw, nh) if (nh, nw) != (h, w) else img
    canvas = np.full((S, S, 3), 114, dtype=np.uint8)
    canvas[top:top + nh, left:left + nw] = res
    out = (canvas.astype(np.float32) / np.float32(255.0)).transpose(2, 0, 1)
    b = np.asarray(boxes, dtype=np.float64).reshape(-1, 4).copy()
    if b.size:
        # albumentations keeps boxes normalised: x / cols survives the resize, then the pad shifts the pixel value
        b[:, [0, 2]] = b[:, [0, 2]] / w * nw + left
        b[:, [1, 3]] = b[:, [1, 3]] / h * nh + top
    return out, b


# ----------------------------------------------------------------------------- image_color_transforms (default.py:420-432,460-461)
# TrainSampleAugmentor's albumentations stage between the warp and the HSV jitter: A.Compose([A.Blur(p=0.01),
# A.MedianBlur(p=0.01), A.ToGray(p=0.01), A.CLAHE(p=0.01)]).  albumentations is neither in the reference tree nor in this image
# and the reference does not pin its version (requirements.txt:26): the GATE below restates the library's draw protocol
# (core/composition.py Compose.__call__: one draw against the Compose's own p = 1; core/transforms_interface.py
# BasicTransform.__call__: one draw per transform against its p; Blur / MedianBlur.get_params: a choice among the odd kernel
# sizes 3..7; CLAHE.get_params: uniform(1, clip_limit = 4.0)) on a generator of the stage's OWN - albumentations >= 1.4 keeps
# a generator per Compose; 1.3.x drew from python's global `random` instead, where these draws (and three more per call of
# the ToFloat / ToTensorV2 Compose) would interleave with DetectionDataset's index draws: a library-version effect the
# reference leaves open, modelled here as the separate stream - and the pixel operations restate OpenCV's published
# algorithms (cv2.blur: box filter,
# BORDER_REFLECT_101; cv2.medianBlur: BORDER_REPLICATE; RGB2GRAY: 15-bit fixed point; CLAHE: imgproc/src/clahe.cpp, on the L
# channel of an 8-bit Lab image).  PARITY UNPINNED, like every other OpenCV restatement in this file - what the reference
# pins is WHERE the stage sits (between random_perspective and augment_hsv) and that it runs at all.
COLOR_BLUR, COLOR_MEDIAN, COLOR_GRAY, COLOR_CLAHE = 1, 2, 4, 8


def color_gate(rnd=random, p: float = 0.01):
    """The draws of one call of the colour Compose: (ops bit mask, blur ksize, median ksize, CLAHE clip limit)."""
    ops, kb, km, clip = 0, 0, 0, 0.0
    rnd.random()                                             # Compose.__call__: need_to_run = random.random() < self.p (p = 1)
    if rnd.random() < p:                                     # Blur(blur_limit=7): ksize in {3, 5, 7}
        ops |= COLOR_BLUR
        kb = int(rnd.choice(list(range(3, 8, 2))))
    if rnd.random() < p:                                     # MedianBlur(blur_limit=7)
        ops |= COLOR_MEDIAN
        km = int(rnd.choice(list(range(3, 8, 2))))
    if rnd.random() < p:                                     # ToGray
        ops |= COLOR_GRAY
    if rnd.random() < p:                                     # CLAHE(clip_limit=4.0 -> (1, 4.0), tile_grid_size=(8, 8))
        ops |= COLOR_CLAHE
        clip = float(rnd.uniform(1, 4.0))
    return ops, kb, km, clip


def _reflect101(i, n):
    i = np.abs(i)
    return np.where(i >= n, 2 * (n - 1) - i, i)


def blur_u8(img: np.ndarray, k: int) -> np.ndarray:
    """cv2.blur(img, (k, k)): normalised box filter, BORDER_REFLECT_101, out = cvRound(sum * (1 / k^2)) (half to even)."""
    h, w = img.shape[:2]
    r = k // 2
    yy = _reflect101(np.arange(-r, h + r), h)
    xx = _reflect101(np.arange(-r, w + r), w)
    p = img[yy][:, xx].astype(np.int64)
    s = np.zeros(img.shape, np.int64)
    for dy in range(k):
        for dx in range(k):
            s += p[dy:dy + h, dx:dx + w]
    return np.clip(np.rint(s.astype(np.float64) * (1.0 / (k * k))), 0, 255).astype(np.uint8)


def median_blur_u8(img: np.ndarray, k: int) -> np.ndarray:
    """cv2.medianBlur(img, k): per-channel median of the k x k window, BORDER_REPLICATE."""
    h, w = img.shape[:2]
    r = k // 2
    p = img[np.clip(np.arange(-r, h + r), 0, h - 1)][:, np.clip(np.arange(-r, w + r), 0, w - 1)]
    win = np.lib.stride_tricks.sliding_window_view(p, (k, k), axis=(0, 1))          # [h][w][c][k][k]
    return np.sort(win.reshape(h, w, img.shape[2], k * k), axis=-1)[..., (k * k) // 2].astype(np.uint8)


def to_gray_u8(img: np.ndarray) -> np.ndarray:
    """albumentations ToGray: cvtColor(RGB2GRAY) then GRAY2RGB - OpenCV's 15-bit weights on channels 0, 1, 2."""
    v = img.astype(np.int64)
    y = (v[..., 0] * 9798 + v[..., 1] * 19235 + v[..., 2] * 3735 + (1 << 14)) >> 15
    return np.repeat(y[..., None], 3, -1).astype(np.uint8)


def lab_tables():
    """Integer tables of the 8-bit RGB <-> Lab round trip (sRGB, D65).  Forward: OpenCV's RGB2Lab_b scheme (color_lab.cpp:
    gamma table x 8, 12-bit matrix, 15-bit cube-root table of 3072 entries).  Inverse: this file's own integer scheme (OpenCV
    4's Lab2RGBinteger could not be restated from its description alone): Q15 f values from per-byte tables, the inverse of f
    in exact integer arithmetic (cube in 64 bits -> Q16), a 12-bit inverse matrix, gamma encoding on a 4096-entry table.
    Returns (gamma u16[256], cbrt u16[3072], C i32[3][3], fy i32[256], dfx i32[256], dfz i32[256], Cinv i32[3][3], enc u8[4096])."""
    i = np.arange(256, dtype=np.float64) / 255.0
    lin = np.where(i <= 0.04045, i / 12.92, ((i + 0.055) / 1.055) ** 2.4)
    gamma = np.clip(np.rint(255.0 * 8 * lin), 0, 65535).astype(np.uint16)
    x = np.arange(3072, dtype=np.float64) / (255.0 * 8)
    cbrt = np.clip(np.rint(32768.0 * np.where(x < 0.008856, x * 7.787 + 16.0 / 116.0, np.cbrt(x))), 0, 65535).astype(np.uint16)
    co = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]])
    white = np.array([0.950456, 1.0, 1.088754])
    C = np.rint(4096.0 * co / white[:, None]).astype(np.int32)
    v = np.arange(256, dtype=np.float64)
    fy = np.rint(32768.0 * (v * 100.0 / 255.0 + 16.0) / 116.0).astype(np.int32)
    dfx = np.rint(32768.0 * (v - 128.0) / 500.0).astype(np.int32)
    dfz = np.rint(32768.0 * (v - 128.0) / 200.0).astype(np.int32)
    Cinv = np.rint(4096.0 * np.linalg.inv(co) * white[None, :]).astype(np.int32)
    u = np.arange(4096, dtype=np.float64) / 4095.0
    enc = np.clip(np.rint(255.0 * np.where(u <= 0.0031308, 12.92 * u, 1.055 * u ** (1 / 2.4) - 0.055)), 0, 255).astype(np.uint8)
    return gamma, cbrt, C, fy, dfx, dfz, Cinv, enc


LAB_T0, LAB_T1, LAB_K = 6780, 4520, 16832        # Q15 6/29 and 4/29; 3 (6/29)^2 / 8 in Q20 (Q15 f -> Q16 value below the knee)


def rgb2lab_u8(img: np.ndarray, T=None) -> np.ndarray:
    gamma, cbrt, C = (T or lab_tables())[:3]
    lin = gamma[img].astype(np.int64)                                                   # [h][w][3]
    f = [cbrt[(lin @ C[r].astype(np.int64) + (1 << 11)) >> 12].astype(np.int64) for r in range(3)]
    Lscale, Lshift = (116 * 255 + 50) // 100, -((16 * 255 * (1 << 15) + 50) // 100)
    L = (Lscale * f[1] + Lshift + (1 << 14)) >> 15
    a = (500 * (f[0] - f[1]) + 128 * (1 << 15) + (1 << 14)) >> 15
    b = (200 * (f[1] - f[2]) + 128 * (1 << 15) + (1 << 14)) >> 15
    return np.clip(np.stack((L, a, b), -1), 0, 255).astype(np.uint8)


def lab2rgb_u8(lab: np.ndarray, T=None) -> np.ndarray:
    _, _, _, fy, dfx, dfz, Cinv, enc = T or lab_tables()
    y = fy[lab[..., 0]].astype(np.int64)
    f = np.clip(np.stack((y + dfx[lab[..., 1]], y, y - dfz[lab[..., 2]]), -1), 0, 65535)
    xyz = np.where(f > LAB_T0, (f * f * f + (1 << 28)) >> 29, np.maximum((LAB_K * (f - LAB_T1) * 16 + (1 << 19)) >> 20, 0))      # Q16
    lin = np.stack([(xyz @ Cinv[r].astype(np.int64) + (1 << 15)) >> 16 for r in range(3)], -1)                             # Q12
    return enc[np.clip(lin, 0, 4095)]


def clahe_plane_u8(src: np.ndarray, clip: float, grid: int = 8) -> np.ndarray:
    """cv::CLAHE::apply on one 8-bit plane (imgproc/src/clahe.cpp): per-tile histogram, clip + redistribute, cumulative LUT,
    bilinear interpolation of the four neighbouring tiles' LUTs (float32, cvRound)."""
    h, w = src.shape
    ph, pw = (grid - h % grid) % grid, (grid - w % grid) % grid
    pad = src[_reflect101(np.arange(h + ph), h)][:, _reflect101(np.arange(w + pw), w)] if (ph or pw) else src
    th, tw = pad.shape[0] // grid, pad.shape[1] // grid
    area = th * tw
    cl = max(int(clip * area / 256), 1)
    scale = np.float32(255.0) / np.float32(area)
    luts = np.zeros((grid, grid, 256), np.uint8)
    for ty in range(grid):
        for tx in range(grid):
            hist = np.bincount(pad[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw].reshape(-1), minlength=256).astype(np.int64)
            clipped = int(np.maximum(hist - cl, 0).sum())
            hist = np.minimum(hist, cl)
            batch, residual = clipped // 256, clipped % 256
            hist += batch
            if residual:
                step = max(256 // residual, 1)
                idx = np.arange(0, 256, step)[:residual]
                hist[idx] += 1
            cum = np.cumsum(hist).astype(np.float32)
            luts[ty, tx] = np.clip(np.rint(cum * scale), 0, 255).astype(np.uint8)

    def axis(n, t):
        f = np.arange(n, dtype=np.float32) * (np.float32(1.0) / np.float32(t)) - np.float32(0.5)
        t1 = np.floor(f).astype(np.int64)
        a = f - t1.astype(np.float32)
        return np.maximum(t1, 0), np.minimum(t1 + 1, grid - 1), a, np.float32(1.0) - a
    ty1, ty2, ya, ya1 = axis(h, th)
    tx1, tx2, xa, xa1 = axis(w, tw)
    v = src.astype(np.int64)
    g = lambda ty, tx: luts[ty[:, None], tx[None, :], v].astype(np.float32)
    res = (g(ty1, tx1) * xa1[None, :] + g(ty1, tx2) * xa[None, :]) * ya1[:, None] + \
          (g(ty2, tx1) * xa1[None, :] + g(ty2, tx2) * xa[None, :]) * ya[:, None]
    return np.clip(np.rint(res), 0, 255).astype(np.uint8)


def clahe_u8(img: np.ndarray, clip: float) -> np.ndarray:
    """albumentations F.clahe on an RGB image: RGB2LAB, CLAHE(clip, 8 x 8) on L, LAB2RGB."""
    T = lab_tables()
    lab = rgb2lab_u8(img, T)
    lab[..., 0] = clahe_plane_u8(lab[..., 0], clip)
    return lab2rgb_u8(lab, T)


def apply_color_ops(img: np.ndarray, ops: int, kb: int, km: int, clip: float) -> np.ndarray:
    """the fired transforms in Compose order"""
    if ops & COLOR_BLUR:
        img = blur_u8(img, kb)
    if ops & COLOR_MEDIAN:
        img = median_blur_u8(img, km)
    if ops & COLOR_GRAY:
        img = to_gray_u8(img)
    if ops & COLOR_CLAHE:
        img = clahe_u8(img, clip)
    return img
