"""TrainSampleAugmentor / mixup / parameter tuples - drop-ins for kod.data.augmentations.default
(kod/data/augmentations/default.py:31-108 AffineParams/HSVParams/AugParams, :400-408 mixup, :411-488
TrainSampleAugmentor; `_target_` of kod/configs/data/augmentations/default.yaml).

Same constructor and call: ``TrainSampleAugmentor(aug_params, rng_seed=51)(sample, border)`` ->
``AugmentedSample(image f32 [3, S, S], bboxes, labels)``.  The generator is consumed in the reference's order
(8 affine draws, 3 HSV draws, 1 flip draw) and the box arithmetic is the same numpy f64; the pixels - warpAffine,
the HSV round trip, the flip, /255 and HWC->CHW - are ONE launch of compose_kernel (csrc/compose.hip) and the image
comes back as a device tensor.  `sample.image` is either the DeviceCanvas MosaicAugmentor returned or a square u8
HWC array (SampleReader output, kod/data/sample_reader.py:102-136), which is treated as a one-tile canvas.
``mixup(a, b)`` blends two such results in the same kernel.  The four p = 0.01 albumentations colour ops
(image_color_transforms, on by default as in the reference) run on the device for the samples whose gate fires
(kodhip_compose_color; the gate's own generator is seeded with rng_seed).
"""
from __future__ import annotations

import numpy as np
import torch

from ... import _lib
from ..device_pipeline import (AffineParams, AugParams, HSVParams, SAMPLE_DESC, _Stager, augment_into, bilinear_table,   # noqa: F401
                               compose)
from ..mosaic import DeviceCanvas, upload_images
from ..types import AugmentedSample


class _Composite(torch.Tensor):
    """f32 [3, S, S] device tensor that remembers the compositing descriptor it was produced from (so mixup can blend
    two samples inside the kernel, bit-equal to the batched path)."""

    @staticmethod
    def wrap(t: torch.Tensor, canvas: DeviceCanvas, desc: np.ndarray, out_size: int):
        r = t.as_subclass(_Composite)
        r._kod = (canvas, desc, out_size)
        return r


class TrainSampleAugmentor(object):
    def __init__(self, aug_params: AugParams, rng_seed: int = 51, device="cuda", albumentations_global_random: bool = False):
        self.aug_params = aug_params
        self.albu13 = bool(albumentations_global_random)        # follow albumentations 1.3.x's draws on python's global generator (host_protocol.augment_into)
        self.rng: np.random.Generator = np.random.default_rng(rng_seed)
        import random
        self.color_rng = random.Random(rng_seed)          # the colour Compose's own stream (host_protocol.color_gate)
        self.device = torch.device(device)
        self._tab = None
        self._stager = None

    def _canvas(self, image) -> DeviceCanvas:
        if isinstance(image, DeviceCanvas):
            return image
        _lib.require_gpu()
        img = np.asarray(image)
        if img.ndim != 3 or img.shape[2] != 3 or img.dtype != np.uint8 or img.shape[0] != img.shape[1]:
            raise ValueError("expected a DeviceCanvas or a square u8 HWC image (SampleReader letter-boxes to S x S)")
        if self._tab is None:
            self._tab = torch.from_numpy(bilinear_table()).to(self.device)
            self._stager = _Stager(self.device)
        pool, offs = upload_images([img], self.device)
        h, w = img.shape[:2]
        return DeviceCanvas(pool, [(int(offs[0]), h, w, (0, 0, w, h), (0, 0))], h, self._tab, self._stager)

    def __call__(self, input_data: AugmentedSample, border=(0, 0)) -> AugmentedSample:
        canvas = self._canvas(input_data.image)
        descs = np.zeros((1, 2), dtype=SAMPLE_DESC)
        canvas.fill(descs[0, 0])
        bb, lb, out = augment_into(descs[0, 0], self.aug_params, self.rng, np.asarray(input_data.bboxes),
                                   np.asarray(input_data.labels), canvas.size, border, color_rng=self.color_rng, albu13=self.albu13)
        mix = np.array([[-1.0, 0.0]], dtype=np.float32)
        img, _, _ = compose(canvas.pool, descs, mix, canvas.tab, out, canvas.stager)
        return AugmentedSample(image=_Composite.wrap(img[0], canvas, descs[0, 0].copy(), out), bboxes=bb, labels=lb)


def mixup(input_data1: AugmentedSample, input_data2: AugmentedSample) -> AugmentedSample:
    """default.py:400-408: r ~ Beta(32, 32) from numpy's global generator; image = im1 * r + im2 * (1 - r)."""
    r = np.random.beta(32.0, 32.0)
    a, b = input_data1.image, input_data2.image
    if not (isinstance(a, _Composite) and isinstance(b, _Composite)):
        raise TypeError("mixup (HIP) blends images produced by TrainSampleAugmentor (they carry their compositing "
                        "descriptors); for whole batches use DeviceTrainPipeline(mixup_prob=...)")
    (ca, da, sa), (cb, db, sb) = a._kod, b._kod
    assert sa == sb, "mixup needs images of one size"
    # both canvases' source images in one pool: rebase the second sample's tile offsets behind the first pool
    descs = np.zeros((1, 2), dtype=SAMPLE_DESC)
    descs[0, 0], descs[0, 1] = da, db
    descs[0, 1]["tile"]["off"] += ca.pool.numel()
    pool = torch.cat((ca.pool, cb.pool))
    mix = np.array([[np.float32(r), np.float32(1 - r)]], dtype=np.float32)
    img, _, _ = compose(pool, descs, mix, ca.tab, sa, ca.stager)
    bboxes = np.concatenate((input_data1.bboxes, input_data2.bboxes), 0)
    labels = np.concatenate((input_data1.labels, input_data2.labels), 0)
    return AugmentedSample(img[0], bboxes, labels)
