"""Seeded synthetic inputs shared by the golden generator, the tests and bench.py.

TEST INFRASTRUCTURE - see oracle/__init__.py.  Pure torch / numpy, no
reference import: the same call reproduces the same inputs on the GPU box.
Target statistics follow SURVEY.md 8(d): n ~ U{1..9} boxes per image, class ~
Zipf(1.01, N=nc) (kod/data/builder.py:110-116), log-uniform box sizes.
"""
from __future__ import annotations

import numpy as np
import torch


def zipf_pmf(nc: int, a: float = 1.01) -> np.ndarray:
    k = np.arange(1, nc + 1, dtype=np.float64)
    p = k ** (-a)
    return p / p.sum()


def targets(B: int, size: int, nc: int, seed: int, nmin: int = 1, nmax: int = 9):
    """List of (boxes f64 [n,4] xyxy px, labels i64 [n])."""
    rng = np.random.default_rng(seed)
    pmf = zipf_pmf(nc)
    out = []
    for _ in range(B):
        n = int(rng.integers(nmin, nmax + 1))
        c = rng.uniform(0, size, (n, 2))
        wh = np.exp(rng.uniform(np.log(8 * size / 640), np.log(400 * size / 640), (n, 2)))
        b = np.concatenate((c - wh / 2, c + wh / 2), 1).clip(0, size - 1)
        ok = ((b[:, 2] - b[:, 0]) > 2) & ((b[:, 3] - b[:, 1]) > 2)
        b = b[ok]
        if b.shape[0] == 0:
            b = np.array([[size * 0.25, size * 0.25, size * 0.75, size * 0.75]])
        lab = rng.choice(nc, size=b.shape[0], p=pmf)
        out.append((torch.from_numpy(b.astype(np.float64)), torch.from_numpy(lab.astype(np.int64))))
    return out


def batch(B: int, size: int, nc: int, seed: int):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 3, size, size, generator=g)
    return x, targets(B, size, nc, seed)


def head_logits(B: int, size: int, nc: int, seed: int, scale: float = 1.0, na: int = 3):
    """Random head outputs [(box,obj,cls)] x 3 levels, contiguous, fp32."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for s in (8, 16, 32):
        h = w = size // s
        out.append((torch.randn(B, na, h, w, 4, generator=g) * scale,
                    torch.randn(B, na, h, w, 1, generator=g) * scale - 2.0,
                    torch.randn(B, na, h, w, nc, generator=g) * scale))
    return out


def _t(boxes, labels):
    return (torch.tensor(boxes, dtype=torch.float64).reshape(-1, 4),
            torch.tensor(labels, dtype=torch.int64))


def assigner_cases():
    """name -> (image size, [(boxes, labels)])."""
    cases = {
        # SURVEY.md Appendix B.1 KAT
        "kat": (640, [_t([[100, 120, 300, 380]], [1]),
                      _t([[400, 50, 460, 130], [10, 10, 30, 40]], [7, 3])]),
        # image without boxes between two with boxes
        "empty_middle": (640, [_t([[50, 60, 90, 130]], [0]), _t([], []), _t([[300, 310, 420, 400]], [2])]),
        # centres on exact integers / half cells / image borders (all five offset blocks)
        "edges": (640, [_t([[8, 8, 24, 24], [0, 0, 16, 16], [600, 600, 639, 639], [316, 316, 324, 324],
                            [60, 60, 68, 68], [12, 20, 28, 28]], [0, 1, 2, 3, 4, 5])]),
        # duplicates (same box twice -> duplicate cells)
        "dups": (640, [_t([[100, 100, 150, 160], [100, 100, 150, 160], [101, 100, 151, 160]], [1, 2, 3])]),
        # nothing at all
        "all_empty": (640, [_t([], []), _t([], [])]),
        # non-640 size
        "s416": (416, targets(4, 416, 10, seed=5)),
        "rand640": (640, targets(8, 640, 10, seed=2023)),
        "mosaic_like": (640, targets(4, 640, 10, seed=9, nmin=10, nmax=36)),
    }
    return cases


def loss_cases():
    """name -> (size, nc, B, targets, pos_weight list | None)."""
    w10 = [float(v) for v in (32567 / np.array([12982, 6918, 2409, 2663, 1837, 1829, 1096, 1009, 941, 883]))]
    return {
        "rand64": (64, 10, 2, targets(2, 64, 10, seed=1), None),
        "rand128_w": (128, 10, 3, targets(3, 128, 10, seed=2, nmin=3, nmax=12), w10),
        "dups128": (128, 10, 2, [_t([[20, 20, 60, 70], [20, 20, 60, 70], [21, 20, 61, 70]], [1, 2, 3]),
                                 _t([[10, 30, 100, 90]], [4])], None),
        "emptyimg128": (128, 10, 2, [_t([], []), _t([[16, 16, 80, 90], [50, 40, 70, 66], [90, 90, 120, 125]],
                                                   [0, 5, 9])], None),
        # tiny boxes only: the hl level gets no match -> NaN box/cls loss (loss.py:96)
        "nan_level128": (128, 10, 1, [_t([[10, 10, 18, 19]], [3])], None),
        "rand256_nc3": (256, 3, 2, targets(2, 256, 3, seed=4), None),
    }


def network_cases():
    """name -> (widen, deepen, nc, B, size, seed)."""
    return {
        "yv5n_64": (0.25, 0.33, 10, 2, 64, 2023),
        "yv5s_64": (0.50, 0.33, 10, 2, 64, 7),
        "yv5s_160": (0.50, 0.33, 10, 2, 160, 2023),
        "yv5s_640": (0.50, 0.33, 10, 2, 640, 2023),
        "yv5s_416": (0.50, 0.33, 10, 4, 416, 2023),       # the reference's default geometry (kod/configs/data/default.yaml:10): 52 / 26 / 13 maps
    }


def decode_cases():
    """name -> (size, nc, B, seed, logit scale)."""
    return {"d64": (64, 10, 2, 21, 1.0), "d128_hot": (128, 10, 2, 22, 3.0), "d96_nc1": (96, 1, 1, 23, 2.0)}


def source_samples(n: int, S: int, seed: int, nc: int = 10):
    """n cached-sample-like tuples (u8 HWC image with longest side S, f64 boxes, i64 labels)."""
    rng = np.random.default_rng(seed)
    ratios = [(4, 3), (3, 4), (1, 1), (3, 2)]
    out = []
    for _ in range(n):
        rw, rh = ratios[int(rng.integers(0, 4))]
        w, h = (S, max(2, int(round(S * rh / rw)))) if rw >= rh else (max(2, int(round(S * rw / rh))), S)
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        k = int(rng.integers(1, 6))
        c = np.stack((rng.uniform(0, w, k), rng.uniform(0, h, k)), 1)
        wh = np.exp(rng.uniform(np.log(max(3.0, S / 40)), np.log(S * 0.6), (k, 2)))
        b = np.concatenate((c - wh / 2, c + wh / 2), 1)
        b[:, [0, 2]] = b[:, [0, 2]].clip(0, w - 1)
        b[:, [1, 3]] = b[:, [1, 3]].clip(0, h - 1)
        ok = ((b[:, 2] - b[:, 0]) > 2) & ((b[:, 3] - b[:, 1]) > 2)
        b = b[ok]
        if b.shape[0] == 0:
            b = np.array([[w * 0.25, h * 0.25, w * 0.75, h * 0.75]])
        out.append((img, b.astype(np.float64), rng.integers(0, nc, b.shape[0]).astype(np.int64)))
    return out


def mosaic_cases():
    return {"m64_a": (64, 1), "m64_b": (64, 2), "m640": (640, 2023), "m416": (416, 11)}


CLASS_COLORS = np.array([[220, 40, 40], [40, 200, 60], [50, 80, 230], [230, 210, 40], [200, 60, 200],
                         [40, 210, 210], [240, 140, 30], [130, 70, 30], [160, 160, 160], [250, 250, 250]], dtype=np.uint8)


def coco_zipf_like(n: int, S: int, seed: int, nc: int = 10):
    """Synthetic detection set: class-coloured rectangles on low-contrast noise, 1-9 objects per image, class
    frequencies ~ Zipf(1.01) (kod/data/builder.py:110-116).  Returns cached-sample tuples (u8 HWC with longest
    side S, boxes f64 xyxy, labels i64)."""
    rng = np.random.default_rng(seed)
    pmf = zipf_pmf(nc)
    out = []
    for _ in range(n):
        rw, rh = [(4, 3), (3, 4), (1, 1), (3, 2)][int(rng.integers(0, 4))]
        w, h = (S, int(round(S * rh / rw))) if rw >= rh else (int(round(S * rw / rh)), S)
        img = rng.integers(90, 140, (h, w, 3), dtype=np.uint8)
        k = int(rng.integers(1, 10))
        boxes, labels = [], []
        for _ in range(k):
            bw, bh = np.exp(rng.uniform(np.log(S / 16), np.log(S / 2.5), 2))
            cx, cy = rng.uniform(bw / 2, w - bw / 2), rng.uniform(bh / 2, h - bh / 2)
            x1, y1, x2, y2 = cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2
            c = int(rng.choice(nc, p=pmf))
            img[int(y1):int(y2), int(x1):int(x2)] = CLASS_COLORS[c % 10]
            boxes.append([x1, y1, x2, y2]); labels.append(c)
        out.append((img, np.array(boxes, dtype=np.float64), np.array(labels, dtype=np.int64)))
    return out


def dataset_info_spec(n: int, n_classes: int, seed: int):
    """Plain-data description of a DatasetInfo (kod/data/cache.py:21-50): class names + per sample (id, [(xyxy, class)]).
    Zipf-like class frequencies; every class occurs; sample ids are decimal strings."""
    rng = np.random.default_rng(seed)
    classes = [f"class_{chr(97 + i)}" for i in range(n_classes)]
    pmf = zipf_pmf(n_classes)
    samples = []
    for i in range(n):
        k = int(rng.integers(1, 5))
        cls = [int(c) for c in rng.choice(n_classes, size=k, p=pmf)]
        if i < n_classes:
            cls[0] = i                                     # every class has at least one image
        tg = []
        for c in cls:
            x1, y1 = rng.uniform(0, 40), rng.uniform(0, 30)
            tg.append(((float(x1), float(y1), float(x1 + rng.uniform(3, 20)), float(y1 + rng.uniform(3, 15))), classes[c]))
        samples.append((str(1000 + i), tg))
    return {"classes": classes, "samples": samples}


# ----------------------------------------------------------------------------- per-sample protocol fixture (tests/golden/protocol.npz)
PROTOCOL_CASES = {
    # name: (mixup_prob, sampler side channel, AugParams overrides)
    "plain": (0.0, False, {}),
    "mix03": (0.3, False, {}),
    "mix10": (1.0, False, {}),
    "rfs03": (0.3, True, {}),                                        # image_repeat_factors + sampler_indices (detection.py:78-80,114-122)
    "rot": (0.0, False, dict(degrees=10.0, shear=5.0, flip=0.0)),    # all matrix factors live; flip_lr_prob = 0: no flip draw
    "nohsv": (0.0, False, dict(hsv=(0.0, 0.0, 0.0))),                # HSVParams.should_aug() False: no HSV draws (default.py:359-364)
    "persp": (0.3, False, dict(degrees=5.0, shear=2.0, perspective=0.0008)),   # cv2.warpPerspective + the boxes' perspective divide (default.py:306-313,257-260)
    # image_color_transforms=True (the reference's default, aug_params.yaml:15): the albumentations colour stage between warp and HSV
    "color": (0.3, False, dict(color=True)),
    # the same under albumentations 1.3.x's draw protocol: every Compose / transform gate draws on python's GLOBAL generator,
    # interleaving with DetectionDataset's index draws (oracle/ref_import.py install_recording, rec.albu13)
    "albu13": (0.3, False, dict(color=True, albu13=True)),
}
PROTOCOL_S, PROTOCOL_POOL, PROTOCOL_N = 64, 12, 64
# generator seed of the colour stage's gate in the 'color' case (the library's own stream; chosen so that each of the four
# p = 0.01 transforms fires at least once within the case's ~85 augmentor calls)
PROTOCOL_COLOR_SEED = 127


def protocol_pool():
    """the synthetic sample pool of the protocol fixture (regenerated identically by the tests)"""
    return source_samples(PROTOCOL_POOL, PROTOCOL_S, seed=9)


def protocol_side_channel():
    """(image_repeat_factors, sampler_indices) of the 'rfs03' case: what a RepeatFactorSampler exposes"""
    rng = np.random.default_rng(4)
    return list(rng.uniform(0.5, 3.0, PROTOCOL_POOL)), [int(i) for i in rng.permutation(PROTOCOL_POOL)]


def sppf_cases():
    """name -> (cin, cout, kernel_sizes, use_conv_first, B, H, W, seed): SPPFBottleneck's three forms (sppf.py:27-83)"""
    return {"k5": (16, 24, 5, True, 2, 12, 10, 31), "k5_9_13": (16, 24, (5, 9, 13), True, 2, 12, 10, 32),
            "k5_9_13_noconv": (8, 24, (5, 9, 13), False, 2, 12, 10, 33),
            # other windows (round 6): a cascade of 3 x 3 pools, parallel pools that are no cascade, one 7 x 7 window without conv1
            "k3": (16, 24, 3, True, 2, 12, 10, 34), "k3_7": (16, 24, (3, 7), True, 2, 12, 10, 35),
            "k7_noconv": (8, 24, (7,), False, 2, 12, 10, 36)}
