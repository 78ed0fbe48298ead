"""MosaicAugmentor - drop-in for kod.data.mosaic.MosaicAugmentor (kod/data/mosaic.py:47-161; `_target_` of
kod/configs/data/default.yaml:14-16).

Same constructor and call: four AugmentedSamples (u8 HWC images, boxes, labels) -> (AugmentedSample, mosaic_border),
same `random.uniform` centre draw, same box shift / filter / clip.  The 2S x 2S canvas itself is NOT pasted on the
host: `image` is a `DeviceCanvas`, the four source images in HBM plus their paste rectangles.  TrainSampleAugmentor
(data/augmentations/default.py) consumes it with ONE launch of compose_kernel (csrc/compose.hip), which gathers
straight from the tiles; `DeviceCanvas.to_uint8()` materialises the reference's canvas when somebody wants to look
at it (identity warp through the same kernel).  Batches should use data/device_pipeline.DeviceTrainPipeline, which
keeps the whole image pool resident and composes a batch per launch; this class is its single-sample form.
"""
from __future__ import annotations

import random
from typing import Sequence, Tuple

import numpy as np
import torch

from .. import _lib
from .device_pipeline import SAMPLE_DESC, _Stager, bilinear_table, compose, mosaic_boxes, mosaic_layout
from .types import AugmentedSample


class DeviceCanvas:
    """A square u8 canvas filled with 114 on which up to four source images (resident in `pool`) are pasted."""

    def __init__(self, pool: torch.Tensor, tiles, size: int, tab: torch.Tensor, stager: _Stager):
        self.pool, self.tiles, self.size, self.tab, self.stager = pool, tiles, size, tab, stager

    @property
    def shape(self):
        return (self.size, self.size, 3)

    def fill(self, desc):
        for t, (off, h, w, a, b) in enumerate(self.tiles):
            d = desc["tile"][t]
            d["off"], d["h"], d["w"] = off, h, w
            d["x1a"], d["y1a"], d["x2a"], d["y2a"] = a
            d["x1b"], d["y1b"] = b

    def to_float(self) -> torch.Tensor:
        """[3, size, size] f32 = canvas / 255 (identity warp: the fixed-point bilinear at integer coordinates is exact)."""
        descs = np.zeros((1, 2), dtype=SAMPLE_DESC)
        self.fill(descs[0, 0])
        descs[0, 0]["im"] = (1.0, 0.0, 0.0, 0.0, 1.0, 0.0)
        descs[0, 0]["canvas"] = self.size
        mix = np.array([[-1.0, 0.0]], dtype=np.float32)
        img, _, _ = compose(self.pool, descs, mix, self.tab, self.size, self.stager)
        return img[0]

    def to_uint8(self) -> torch.Tensor:
        """The reference's img4: [size, size, 3] u8 on the device."""
        return (self.to_float() * 255.0).round().to(torch.uint8).permute(1, 2, 0).contiguous()


def upload_images(images: Sequence[np.ndarray], device):
    """u8 HWC images -> one flat device buffer + byte offsets."""
    flats = [np.ascontiguousarray(im, dtype=np.uint8).reshape(-1) for im in images]
    offs = np.concatenate(([0], np.cumsum([f.size for f in flats])[:-1])).astype(np.int64)
    # (+ 8 bytes of slack: the compositing kernel fetches a pixel's three channel bytes with one 4-byte load)
    host = torch.from_numpy(np.concatenate(flats + [np.zeros(8, np.uint8)])).pin_memory()
    return host.to(device, non_blocking=True), offs


class MosaicAugmentor(object):
    def __init__(self, target_image_size: int, device="cuda"):
        self.target_size = target_image_size
        self.device = torch.device(device)
        self._tab = None
        self._stager = None

    def __call__(self, input_data: Sequence[AugmentedSample]) -> Tuple[AugmentedSample, Tuple[int, int]]:
        assert len(input_data) == 4, "mosaic input_data must be a list containing 4 images"
        _lib.require_gpu()
        S = self.target_size
        mosaic_border = (-S // 2, -S // 2)
        yc, xc = (int(random.uniform(-x, 2 * S + x)) for x in mosaic_border)          # mosaic.py:58-62
        shapes = [d.image.shape[:2] for d in input_data]
        rects = mosaic_layout(shapes, xc, yc, S)
        bb, lb = mosaic_boxes([(None, np.asarray(d.bboxes), np.asarray(d.labels)) for d in input_data], rects, S)
        if self._tab is None:
            self._tab = torch.from_numpy(bilinear_table()).to(self.device)
            self._stager = _Stager(self.device)
        pool, offs = upload_images([d.image for d in input_data], self.device)
        tiles = [(int(offs[i]), shapes[i][0], shapes[i][1], rects[i][0], rects[i][1]) for i in range(4)]
        canvas = DeviceCanvas(pool, tiles, 2 * S, self._tab, self._stager)
        return AugmentedSample(image=canvas, bboxes=bb, labels=lb), mosaic_border
