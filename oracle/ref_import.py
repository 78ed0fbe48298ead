"""Import the real reference (``kod``) in the BUILD container to pin the oracle.

TEST INFRASTRUCTURE - see oracle/__init__.py.  Used only by
``oracle/gen_golden.py`` (fixture generation) and never on the GPU box, where
``/root/reference`` does not exist.

The reference depends on wheels that are absent here (torchvision, absl, cv2,
albumentations).  Only three torchvision symbols take part in arithmetic on
the hot path; they are restated below with torchvision 0.15.2 semantics
(the version pinned by the reference's requirements.txt:17-18).  ``absl``,
``cv2`` and ``albumentations`` are import-only placeholders: any reference
function that really calls into them (warpAffine, cvtColor, ...) is NOT
reachable through ``install()`` and stays "parity unpinned".

``install_recording()`` (used by ``gen_golden.gen_protocol``) fills those two
placeholders with RECORDING stand-ins so that the reference's real
``DetectionDataset.__getitem__`` / ``TrainSampleAugmentor.__call__`` can run
end to end: every ``cv2`` call is logged with its arguments (the affine
matrices, output sizes, HSV look-up tables) and the pixel work itself is
delegated to the oracle's own restatements (``oracle/datapath.py``).  That
pins everything AROUND the pixels - index draws, mosaic geometry, RNG draw
order, matrices, LUTs, flips, the mixup ratio, boxes and labels, the order
in which the stages are composed - to the reference; the pixel arithmetic
of OpenCV itself stays unpinned.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = os.environ.get("KOD_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "kod"))


class _ConvNormAct(nn.Sequential):
    """torchvision.ops.misc.Conv2dNormActivation (0.15.2) structure: children 0,1,2."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=None,
                 groups=1, norm_layer=nn.BatchNorm2d, activation_layer=nn.ReLU, dilation=1,
                 inplace=True, bias=None):
        if padding is None:
            padding = (kernel_size - 1) // 2 * dilation
        if bias is None:
            bias = norm_layer is None
        layers = [nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding,
                            dilation=dilation, groups=groups, bias=bias)]
        if norm_layer is not None:
            layers.append(norm_layer(out_channels))
        if activation_layer is not None:
            kw = {} if inplace is None else {"inplace": inplace}
            layers.append(activation_layer(**kw))
        super().__init__(*layers)
        self.out_channels = out_channels


def _box_convert(boxes, in_fmt, out_fmt):
    if in_fmt == out_fmt:
        return boxes.clone()
    if (in_fmt, out_fmt) == ("xyxy", "cxcywh"):
        x1, y1, x2, y2 = boxes.unbind(-1)
        return torch.stack(((x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1), -1)
    if (in_fmt, out_fmt) == ("cxcywh", "xyxy"):
        cx, cy, w, h = boxes.unbind(-1)
        return torch.stack((cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h), -1)
    raise NotImplementedError((in_fmt, out_fmt))


def _nms(boxes, scores, iou_threshold):
    order = torch.argsort(scores, descending=True, stable=True)
    b = boxes[order]
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    dead = torch.zeros(b.shape[0], dtype=torch.bool)
    keep = []
    for i in range(b.shape[0]):
        if dead[i]:
            continue
        keep.append(i)
        lt = torch.max(b[i, :2], b[i + 1:, :2])
        rb = torch.min(b[i, 2:], b[i + 1:, 2:])
        wh = (rb - lt).clamp(min=0)
        inter = wh[:, 0] * wh[:, 1]
        dead[i + 1:] |= inter / (area[i] + area[i + 1:] - inter) > iou_threshold
    return order[torch.tensor(keep, dtype=torch.long)]


class _Anything:
    def __getattr__(self, name):
        return _Anything()

    def __call__(self, *a, **k):
        return _Anything()


def _placeholder(name: str, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    def _missing(attr):
        if attr.startswith("__"):
            raise AttributeError(attr)
        return _Anything()

    m.__getattr__ = _missing                      # type: ignore[attr-defined]
    return m


class ToTensorV2:
    """albumentations.pytorch.ToTensorV2: torch.from_numpy(img.transpose(2, 0, 1)).  The reference binds this name at
    import (`from albumentations.pytorch import ToTensorV2`, kod/data/augmentations/default.py), so the stand-in has to be
    in place BEFORE the first reference import, whichever generator runs first - install() puts it there."""

    draw = None          # install_recording: the library-generation switch's draw hook (albumentations 1.3.x: one draw per call)

    def __init__(self, *a, **k):
        pass

    def __call__(self, **data):
        if ToTensorV2.draw is not None:
            ToTensorV2.draw()
        return dict(data, image=torch.from_numpy(data["image"].transpose(2, 0, 1)))


_installed = False


def install():
    """Register the shim modules and put the reference on sys.path (idempotent)."""
    global _installed
    if _installed or "kod" in sys.modules:
        return
    _installed = True
    sys.dont_write_bytecode = True
    tv = types.ModuleType("torchvision")
    ops = types.ModuleType("torchvision.ops")
    misc = types.ModuleType("torchvision.ops.misc")
    misc.Conv2dNormActivation = _ConvNormAct
    ops.misc, ops.box_convert, ops.nms = misc, _box_convert, _nms
    tv.ops = ops
    sys.modules.update({"torchvision": tv, "torchvision.ops": ops, "torchvision.ops.misc": misc})
    absl = _placeholder("absl")
    absl_logging = _placeholder("absl.logging", DEBUG=0, INFO=1,
                                set_verbosity=lambda *_: None, info=lambda *a, **k: None,
                                warning=lambda *a, **k: None, debug=lambda *a, **k: None)
    absl.logging = absl_logging
    sys.modules.update({"absl": absl, "absl.logging": absl_logging})
    for name in ("cv2", "albumentations", "albumentations.pytorch", "albumentations.core",
                 "albumentations.core.composition"):
        sys.modules[name] = _placeholder(name, TransformsSeqType=list)
    sys.modules["albumentations.pytorch"].ToTensorV2 = ToTensorV2
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def ref(module: str):
    install()
    return importlib.import_module(module)


# ----------------------------------------------------------------------------- recording stand-ins (gen_protocol)
class Recorder:
    """Event log of one run of the reference's data path: (name, payload) in call order."""

    def __init__(self):
        self.events = []

    def __call__(self, name, **payload):
        self.events.append((name, payload))

    def take(self):
        ev, self.events = self.events, []
        return ev


def install_recording(rec: Recorder):
    """cv2 / albumentations stand-ins that log what the reference asks of them and hand the pixel work to the oracle's
    restatements.  Call after install(); idempotent per recorder."""
    import zlib

    import numpy as np

    install()
    from oracle import datapath as DP

    cv2 = sys.modules["cv2"]
    cv2.INTER_LINEAR, cv2.BORDER_CONSTANT, cv2.COLOR_BGR2HSV, cv2.COLOR_HSV2BGR = 1, 0, 40, 54

    def getRotationMatrix2D(angle, center, scale):
        # OpenCV documentation, cv::getRotationMatrix2D: alpha = scale cos(angle), beta = scale sin(angle) (degrees),
        # [[alpha, beta, (1 - alpha) cx - beta cy], [-beta, alpha, beta cx + (1 - alpha) cy]]
        import math
        a = math.radians(angle)
        al, be = scale * math.cos(a), scale * math.sin(a)
        cx, cy = center
        return np.array([[al, be, (1 - al) * cx - be * cy], [-be, al, be * cx + (1 - al) * cy]], dtype=np.float64)

    def warpAffine(im, M, dsize, borderValue=(0, 0, 0), flags=1, borderMode=0):
        assert flags == cv2.INTER_LINEAR and borderMode == cv2.BORDER_CONSTANT and tuple(borderValue) == (114, 114, 114)
        rec("warpAffine", M=np.array(M, dtype=np.float64), dsize=tuple(int(v) for v in dsize),
            src_shape=tuple(im.shape), src_crc=zlib.crc32(np.ascontiguousarray(im).tobytes()))
        return DP.warp_affine_u8(im, np.asarray(M, dtype=np.float64), int(dsize[0]), int(dsize[1]), 114)

    def warpPerspective(im, M, dsize, borderValue=(0, 0, 0), flags=1, borderMode=0):
        assert flags == cv2.INTER_LINEAR and borderMode == cv2.BORDER_CONSTANT and tuple(borderValue) == (114, 114, 114)
        rec("warpPerspective", M=np.array(M, dtype=np.float64), dsize=tuple(int(v) for v in dsize),
            src_shape=tuple(im.shape), src_crc=zlib.crc32(np.ascontiguousarray(im).tobytes()))
        return DP.warp_perspective_u8(im, np.asarray(M, dtype=np.float64), int(dsize[0]), int(dsize[1]), 114)

    def cvtColor(img, code):
        rec("cvtColor", code=int(code))
        return DP.bgr2hsv_u8(img) if code == cv2.COLOR_BGR2HSV else DP.hsv2bgr_u8(img)

    def LUT(ch, lut):
        rec("LUT", lut=np.array(lut))
        return np.asarray(lut)[ch]

    cv2.getRotationMatrix2D, cv2.warpAffine, cv2.warpPerspective, cv2.cvtColor, cv2.LUT = \
        getRotationMatrix2D, warpAffine, warpPerspective, cvtColor, LUT
    cv2.split = lambda img: tuple(img[..., i] for i in range(img.shape[-1]))
    cv2.merge = lambda chs: np.stack(chs, -1)

    A = sys.modules["albumentations"]

    # Library generation (the reference does not pin albumentations, requirements.txt:26).  rec.albu13 = False (default):
    # albumentations >= 1.4 - every Compose draws on a generator of its own; only the colour stage's draws are modelled
    # (rec.color_rng), the ToFloat / ToTensorV2 Compose consumes nothing anyone else sees.  rec.albu13 = True: albumentations
    # 1.3.x - Compose.__call__ and every BasicTransform.__call__ draw `random.random()` on python's GLOBAL generator (one draw
    # per Compose call + one per transform), so they interleave with DetectionDataset's index draws.
    import random as _globalrandom

    def _legacy_draw():
        if getattr(rec, "albu13", False):
            rec("albu13_draw")
            _globalrandom.random()
    ToTensorV2.draw = staticmethod(_legacy_draw)

    class Compose:                       # albumentations.Compose: transforms applied in order to data["image"]
        def __init__(self, transforms, *a, **k):
            self.transforms = list(transforms)

        def __call__(self, **data):
            if not any(getattr(t, "colour_stage", False) for t in self.transforms):
                _legacy_draw()           # (the colour Compose makes its Compose-level draw in ColourAwareCompose)
            for t in self.transforms:
                data = t(**data)
            return data

    class ToFloat:                       # albumentations ToFloat: img.astype("float32") / max_value
        def __init__(self, max_value=None, **k):
            self.max_value = max_value

        def __call__(self, **data):
            _legacy_draw()
            rec("ToFloat", max_value=self.max_value)
            return dict(data, image=data["image"].astype("float32") / self.max_value)

    # The colour stage (default.py:420-432: A.Compose([A.Blur(p=0.01), A.MedianBlur(p=0.01), A.ToGray(p=0.01),
    # A.CLAHE(p=0.01)])): each stand-in takes the p the REFERENCE passes, draws against it on the stage's own generator
    # (rec.color_rng, see oracle/datapath.color_gate for the protocol and why it is a separate stream), logs what fired
    # and hands the pixels to the oracle's restatement.  A Compose that holds such transforms makes the Compose-level draw.
    import random as _pyrandom

    def _colour_op(name, bit, draw_params, apply):
        class Op:
            colour_stage = True

            def __init__(self, *a, p=0.5, **k):
                self.p = p

            def __call__(self, **data):
                g = _globalrandom if getattr(rec, "albu13", False) else getattr(rec, "color_rng", None)
                assert g is not None, "set Recorder.color_rng before running a configuration with image_color_transforms=True"
                if g.random() < self.p:
                    params = draw_params(g)
                    rec("color", op=name, bit=bit, **params)
                    return dict(data, image=apply(data["image"], **params))
                return data
        Op.__name__ = name
        return Op

    class ColourAwareCompose(Compose):
        def __call__(self, **data):
            if any(getattr(t, "colour_stage", False) for t in self.transforms):
                rec("color_stage", n=len(self.transforms))
                (_globalrandom if getattr(rec, "albu13", False) else rec.color_rng).random()   # Compose.__call__: need_to_run = random() < self.p (p = 1)
            return super().__call__(**data)

    A.Compose, A.ToFloat = ColourAwareCompose, ToFloat
    odd = list(range(3, 8, 2))
    A.Blur = _colour_op("Blur", DP.COLOR_BLUR, lambda g: dict(ksize=int(g.choice(odd))), lambda im, ksize: DP.blur_u8(im, ksize))
    A.MedianBlur = _colour_op("MedianBlur", DP.COLOR_MEDIAN, lambda g: dict(ksize=int(g.choice(odd))),
                              lambda im, ksize: DP.median_blur_u8(im, ksize))
    A.ToGray = _colour_op("ToGray", DP.COLOR_GRAY, lambda g: {}, lambda im: DP.to_gray_u8(im))
    A.CLAHE = _colour_op("CLAHE", DP.COLOR_CLAHE, lambda g: dict(clip_limit=float(g.uniform(1, 4.0))),
                         lambda im, clip_limit: DP.clahe_u8(im, clip_limit))
    del _pyrandom
