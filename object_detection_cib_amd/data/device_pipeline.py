"""Device-side training data path: the host keeps the reference's per-sample protocol (index choice, RNG
draw order, box arithmetic in numpy f64) and hands ONE descriptor per sample to the HIP compositing kernel
(csrc/compose.hip), which produces the batch directly in HBM - the 2S x 2S mosaic canvas, the warped u8
image and the fp32 CHW tensor of the reference pipeline are never materialised on the host.

Mirrors (same draw order, same arithmetic for boxes):
  DetectionDataset.__getitem__          kod/data/detection.py:102-156
  MosaicAugmentor.__call__              kod/data/mosaic.py:51-161
  TrainSampleAugmentor.__call__         kod/data/augmentations/default.py:440-488
  random_perspective / boxes / flip     kod/data/augmentations/default.py:111-351,386-397
  mixup                                 kod/data/augmentations/default.py:400-408
The four p=0.01 albumentations colour ops (Blur / MedianBlur / ToGray / CLAHE, default.py:420-432,460-461:
`image_color_transforms`, on by default like in the reference) run on the device for the samples whose gate fired
(csrc/compose.hip kodhip_compose_color; gate: host_protocol.color_gate).
"""
from __future__ import annotations

import math
import random as _random
from typing import List, NamedTuple, Sequence, Tuple

import numpy as np
import torch

from .. import _lib
from .detection import DetectionTarget


from .host_protocol import (AffineParams, HSVParams, AugParams, SAMPLE_DESC, HostProtocol, PackedTargets, pack_targets,   # noqa: F401
                            mosaic_layout, mosaic_boxes, affine_matrix, invert_affine, affine_boxes, augment_into, _box_candidates,
                            color_gate)


def color_table() -> np.ndarray:
    """Integer tables of the colour stage's 8-bit RGB <-> Lab round trip (CLAHE runs on L; csrc/compose.hip LabTab), as one
    byte array: gamma u16[256] | cube root u16[3072] | forward matrix i32[9] | inverse matrix i32[9] | fy, dfx, dfz i32[256]
    each | gamma encoding u8[4096].  Forward: OpenCV's RGB2Lab_b scheme (color_lab.cpp: sRGB gamma table x 8, Q12 matrix over
    the D65 white point, Q15 cube-root table); inverse: f values per byte in Q15, the inverse of f in exact integer
    arithmetic in the kernel, Q12 inverse matrix, gamma encoding by table (OpenCV 4's integer Lab2RGB is not restated;
    parity unpinned).  Same numbers as oracle/datapath.lab_tables()."""
    i = np.arange(256, dtype=np.float64) / 255.0
    gamma = np.clip(np.rint(255.0 * 8 * np.where(i <= 0.04045, i / 12.92, ((i + 0.055) / 1.055) ** 2.4)), 0, 65535).astype(np.uint16)
    x = np.arange(3072, dtype=np.float64) / (255.0 * 8)
    cbrt = np.clip(np.rint(32768.0 * np.where(x < 0.008856, x * 7.787 + 16.0 / 116.0, np.cbrt(x))), 0, 65535).astype(np.uint16)
    co = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]])
    white = np.array([0.950456, 1.0, 1.088754])
    C = np.rint(4096.0 * co / white[:, None]).astype(np.int32)
    Cinv = np.rint(4096.0 * np.linalg.inv(co) * white[None, :]).astype(np.int32)
    v = np.arange(256, dtype=np.float64)
    fy = np.rint(32768.0 * (v * 100.0 / 255.0 + 16.0) / 116.0).astype(np.int32)
    dfx = np.rint(32768.0 * (v - 128.0) / 500.0).astype(np.int32)
    dfz = np.rint(32768.0 * (v - 128.0) / 200.0).astype(np.int32)
    u = np.arange(4096, dtype=np.float64) / 4095.0
    enc = np.clip(np.rint(255.0 * np.where(u <= 0.0031308, 12.92 * u, 1.055 * u ** (1 / 2.4) - 0.055)), 0, 255).astype(np.uint8)
    parts = [gamma, cbrt, C.reshape(-1), Cinv.reshape(-1), fy, dfx, dfz, enc]
    out = np.concatenate([np.ascontiguousarray(a).view(np.uint8).reshape(-1) for a in parts])
    assert out.size == 9800 + 4096
    return out


_COLOR_TABS = {}


def _color_stage(pool: torch.Tensor, descs: np.ndarray, mix: np.ndarray, S: int):
    """Entries (2 * sample + slot) of descs [B][2] whose colour gate fired, and the u8 scratch images the stage leaves its
    results in; sets every such descriptor's `pre` (so call BEFORE descs is uploaded).  Returns [] or
    [(entry, ops, blur_k, median_k, clip, out, tmp, luts)]."""
    hit = [(b, s) for b, s in zip(*np.nonzero(descs["color"])) if s == 0 or mix[b, 0] >= 0]
    if not hit:
        return []
    scratch = torch.empty((len(hit), 2, S * S * 3 + 64 * 256), dtype=torch.uint8, device=pool.device)
    jobs = []
    for k, (b, s) in enumerate(hit):
        d = descs[b, s]
        out, tmp = scratch[k, 0], scratch[k, 1]
        d["pre"] = out.data_ptr()
        jobs.append((int(b) * 2 + int(s), int(d["color"]), int(d["blur_k"]), int(d["median_k"]), float(d["clahe_clip"]),
                     out, tmp, tmp[S * S * 3:]))
    return jobs


def bilinear_table() -> np.ndarray:
    """The constant tables of the compositing kernel as one int16 array: OpenCV's fixed-point INTER_LINEAR table
    (imgwarp.cpp initInterTab2D, fixpt): [32*32][4] int16 weights (y0x0, y0x1, y1x0, y1x1) = saturate_cast<short>(wy*wx*32768) -
    all products are exact multiples of 32, so every entry sums to 32768 except (0,0), where 32768 saturates to 32767
    (OpenCV's sum fix-up for that entry lands outside the 2x2 block and is overwritten, so it stays 32767) - followed by the
    two division tables of cvtColor's RGB2HSV_b (color_hsv.simd.hpp: sdiv_table[i] = cvRound((255 << 12) / (1. * i)),
    hdiv_table180[i] = cvRound((180 << 12) / (6. * i)), entry 0 = 0; int32 [256] each, stored as int16 pairs)."""
    n = 32
    t1 = np.stack((1.0 - np.arange(n) / n, np.arange(n) / n), 1).astype(np.float32)
    tab = np.zeros((n, n, 4), dtype=np.int16)
    for i in range(n):
        for j in range(n):
            f = (t1[i][:, None] * t1[j][None, :]).astype(np.float32).reshape(-1)
            tab[i, j] = np.clip(np.rint(f.astype(np.float64) * 32768), -32768, 32767).astype(np.int16)
    i = np.arange(1, 256, dtype=np.float64)
    sdiv = np.concatenate(([0], np.rint((255 << 12) / i))).astype(np.int32)
    hdiv = np.concatenate(([0], np.rint((180 << 12) / (6.0 * i)))).astype(np.int32)
    return np.concatenate((tab.reshape(-1), sdiv.view(np.int16), hdiv.view(np.int16)))


class _Stager:
    """Host -> device upload of small per-batch tables without stalling the host on the stream: a ring of pinned
    buffers + non-blocking copies (a pageable-memory copy waits for everything queued on the stream before it, which
    serialises the data pipeline behind the previous training step); a slot is reused only after its copy ran."""

    def __init__(self, device, depth: int = 8):
        self.device, self.depth = device, depth
        self.slots, self.events, self.i = [None] * depth, [None] * depth, 0

    def upload(self, arr: np.ndarray) -> torch.Tensor:
        raw = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
        size = (raw.size + 15) // 16 * 16              # (kodhip_pull_from_host moves 16-byte pieces)
        k = self.i
        self.i = (self.i + 1) % self.depth
        if self.events[k] is not None:
            self.events[k].synchronize()
        if self.slots[k] is None or self.slots[k].numel() < size:
            self.slots[k] = torch.empty(max(size, 1024), dtype=torch.uint8).pin_memory()
        host = self.slots[k][:size]
        host.numpy()[:raw.size] = raw
        dev = torch.empty(size, dtype=torch.uint8, device=self.device)
        # a kernel pulling the pinned bytes through their device mapping: an async copy queued on the stream costs the stream
        # ~0.15 ms whatever its size on this stack (tools/loop_parts.py)
        _lib.check(_lib.lib().kodhip_pull_from_host(dev.data_ptr(), host.data_ptr(), size, torch.cuda.current_stream().cuda_stream),
                   "pull_from_host")
        ev = torch.cuda.Event()
        ev.record()
        self.events[k] = ev
        return dev[:raw.size]


def compose(pool: torch.Tensor, descs: np.ndarray, mix: np.ndarray, tab: torch.Tensor, S: int, stager: "_Stager",
            out_f32: bool = True, out_pairs: bool = False, pairs_out: torch.Tensor = None):
    """One launch of compose_kernel for descs [B][2] / mix [B][2]; returns (f32 [B,3,S,S] | None, pairs | None).
    pairs_out: write the pixel pairs into this tensor (the network's own input buffer, Engine.image_buffer) instead of a new one."""
    B = descs.shape[0]
    jobs = _color_stage(pool, descs, mix, S)                 # image_color_transforms: the ~4 % of samples whose gate fired
    d_dev = stager.upload(descs)
    m_dev = stager.upload(mix).view(torch.float32)
    if jobs:
        ctab = _COLOR_TABS.get(pool.device)
        if ctab is None:
            ctab = _COLOR_TABS[pool.device] = torch.from_numpy(color_table()).to(pool.device)
        for entry, ops, kb, km, clip, out, tmp, luts in jobs:
            _lib.check(_lib.lib().kodhip_compose_color(pool.data_ptr(), d_dev.data_ptr(), tab.data_ptr(), ctab.data_ptr(), entry,
                                                       ops, kb, km, clip, out.data_ptr(), tmp.data_ptr(), luts.data_ptr(), S,
                                                       torch.cuda.current_stream().cuda_stream), "compose_color")
    img = torch.empty((B, 3, S, S), dtype=torch.float32, device=pool.device) if out_f32 else None
    pairs = None
    if out_pairs:
        if pairs_out is not None:
            assert tuple(pairs_out.shape) == (B, S, S // 2, 8) and pairs_out.dtype == torch.bfloat16 and pairs_out.is_contiguous()
            pairs = pairs_out
        else:
            pairs = torch.empty((B, S, S // 2, 8), dtype=torch.bfloat16, device=pool.device)
    _lib.check(_lib.lib().kodhip_compose_batch(pool.data_ptr(), d_dev.data_ptr(), m_dev.data_ptr(), tab.data_ptr(),
                                               img.data_ptr() if out_f32 else None,
                                               pairs.data_ptr() if out_pairs else None, B, S,
                                               torch.cuda.current_stream().cuda_stream), "compose_batch")
    return img, pairs, (d_dev, m_dev, jobs)


class ImagePool:
    """RAM-cache analogue resident in HBM: all resized source images (u8 HWC, longest side <= S) in one buffer."""

    def __init__(self, images: Sequence[np.ndarray], device):
        self.shapes = [(im.shape[0], im.shape[1]) for im in images]
        sizes = [im.shape[0] * im.shape[1] * 3 for im in images]
        self.offsets = np.concatenate(([0], np.cumsum(sizes)[:-1])).astype(np.int64)
        # (+ 8 bytes of slack: the compositing kernel fetches a pixel's three channel bytes with one 4-byte load)
        flat = np.concatenate([np.ascontiguousarray(im, dtype=np.uint8).reshape(-1) for im in images] + [np.zeros(8, np.uint8)])
        self.data = torch.from_numpy(flat).to(device)


class DeviceTrainPipeline:
    """Produces training batches on the GPU with the reference's sampling / augmentation protocol."""

    def __init__(self, images: Sequence[np.ndarray], boxes: Sequence[np.ndarray], labels: Sequence[np.ndarray],
                 target_image_size: int, device, aug_params: AugParams = AugParams(), mixup_prob: float = 0.0,
                 rng_seed: int = 51, image_repeat_factors=None, sampler_indices=None, albumentations_global_random: bool = False):
        _lib.require_gpu()
        assert desc_bytes() == SAMPLE_DESC.itemsize, (desc_bytes(), SAMPLE_DESC.itemsize)
        self.S = int(target_image_size)
        self.device = torch.device(device)
        self.pool = ImagePool(images, self.device)
        self.host = HostProtocol(self.pool.shapes, self.pool.offsets, boxes, labels, target_image_size, aug_params,
                                 mixup_prob, rng_seed, image_repeat_factors, sampler_indices, albumentations_global_random)
        self.tab = torch.from_numpy(bilinear_table()).to(self.device)
        self._stager = _Stager(self.device)

    # (the protocol state lives in self.host; these stay assignable for tests that reset the augmentor's generator)
    @property
    def rng(self):
        return self.host.rng

    @rng.setter
    def rng(self, g):
        self.host.rng = g

    def host_args(self) -> dict:
        """Constructor arguments of an equivalent HostProtocol (picklable): what data/producer.py starts its worker with."""
        h = self.host
        return dict(shapes=h.shapes, offsets=h.offsets, boxes=h.boxes, labels=h.labels, target_image_size=h.S,
                    aug_params=h.aug, mixup_prob=h.mixup_prob, image_repeat_factors=h.weights, sampler_indices=h.sampler_indices,
                    albumentations_global_random=h.albu13)

    def compose_host_batch(self, descs: np.ndarray, mix: np.ndarray, out_f32: bool = True, out_pairs: bool = False, pairs_out=None):
        """The device half alone: descriptors (from this process or from a producer process) -> (f32 | None, pairs | None).
        pairs_out: the tensor to composite into (e.g. the network's input buffer: no copy between pipeline and step)."""
        img, pairs, self._keep = compose(self.pool.data, descs, mix, self.tab, self.S, self._stager, out_f32, out_pairs, pairs_out)
        return img, pairs

    def make_batch(self, batch_indices: Sequence[int], out_f32: bool = True, out_pairs: bool = False):
        """Returns (images f32 [B,3,S,S] or None, pairs bf16 [B,S,S/2,8] or None, tuple of DetectionTarget)."""
        descs, mix, per_sample = self.host.batch(batch_indices)
        targets = tuple(DetectionTarget(torch.from_numpy(bb), torch.from_numpy(lb)) for bb, lb in per_sample)
        img, pairs = self.compose_host_batch(descs, mix, out_f32, out_pairs)
        return img, pairs, targets


def desc_bytes() -> int:
    return _lib.lib().kodhip_compose_desc_bytes()


# ------------------------------------------------------------------------------------------- validation batches
VAL_DESC = np.dtype([("off", "<i8"), ("h", "<i4"), ("w", "<i4"), ("nh", "<i4"), ("nw", "<i4"), ("top", "<i4"),
                     ("left", "<i4"), ("scale_x", "<f8"), ("scale_y", "<f8")], align=True)


def _py3round(v: float) -> int:
    """albumentations' py3round (geometric/functional.py): exact halves round away from zero."""
    if abs(round(v) - v) == 0.5:
        return int(2.0 * round(v / 2.0))
    return int(round(v))


def letterbox_geometry(h: int, w: int, S: int):
    """(new_h, new_w, pad_top, pad_left) of A.LongestMaxSize(S) followed by A.PadIfNeeded(S, S), centre position
    (kod/data/sample_reader.py:16-40)."""
    scale = S / float(max(w, h))
    nh, nw = (_py3round(h * scale), _py3round(w * scale)) if scale != 1.0 else (h, w)
    top = int((S - nh) / 2.0) if nh < S else 0
    left = int((S - nw) / 2.0) if nw < S else 0
    return nh, nw, top, left


class DeviceValPipeline:
    """Validation batches on the GPU: SampleReader(letter_box=True) + ValidationSampleAugmentor
    (kod/data/sample_reader.py:102-136, kod/data/augmentations/albu.py:91-119) for a pool of ORIGINAL-size u8
    images resident in HBM - resize (cv2 INTER_LINEAR fixed point), pad 114, /255, CHW in one launch."""

    def __init__(self, images: Sequence[np.ndarray], boxes: Sequence[np.ndarray], labels: Sequence[np.ndarray],
                 target_image_size: int, device):
        _lib.require_gpu()
        assert _lib.lib().kodhip_val_prep_desc_bytes() == VAL_DESC.itemsize
        self.S = int(target_image_size)
        self.device = torch.device(device)
        self.pool = ImagePool(images, self.device)
        self.boxes, self.labels = list(boxes), list(labels)
        self._stager = _Stager(self.device)

    def make_batch(self, batch_indices: Sequence[int], out_f32: bool = True, out_pairs: bool = False):
        """Returns (images f32 [B,3,S,S] or None, pairs bf16 [B,S,S/2,8] or None, tuple of DetectionTarget)."""
        B, S = len(batch_indices), self.S
        descs = np.zeros(B, dtype=VAL_DESC)
        targets = []
        for k, i in enumerate(batch_indices):
            h, w = self.pool.shapes[i]
            nh, nw, top, left = letterbox_geometry(h, w, S)
            d = descs[k]
            d["off"], d["h"], d["w"], d["nh"], d["nw"], d["top"], d["left"] = self.pool.offsets[i], h, w, nh, nw, top, left
            d["scale_x"], d["scale_y"] = 1.0 / (nw / float(w)), 1.0 / (nh / float(h))     # OpenCV: 1 / inv_scale
            bb = np.asarray(self.boxes[i], dtype=np.float64).reshape(-1, 4).copy()
            if bb.size:          # albumentations keeps boxes normalised through the resize; the pad shifts pixels
                bb[:, [0, 2]] = bb[:, [0, 2]] / w * nw + left
                bb[:, [1, 3]] = bb[:, [1, 3]] / h * nh + top
            targets.append(DetectionTarget(torch.from_numpy(bb), torch.from_numpy(np.asarray(self.labels[i], dtype=np.int64))))
        d_dev = self._stager.upload(descs)
        img = torch.empty((B, 3, S, S), dtype=torch.float32, device=self.device) if out_f32 else None
        pairs = torch.empty((B, S, S // 2, 8), dtype=torch.bfloat16, device=self.device) if out_pairs else None
        _lib.check(_lib.lib().kodhip_val_prep_batch(self.pool.data.data_ptr(), d_dev.data_ptr(),
                                                    img.data_ptr() if out_f32 else None,
                                                    pairs.data_ptr() if out_pairs else None, B, S,
                                                    torch.cuda.current_stream().cuda_stream), "val_prep_batch")
        self._keep = d_dev
        return img, pairs, tuple(targets)
