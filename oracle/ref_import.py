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
reachable through this shim and stays "parity unpinned".
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


def install():
    """Register the shim modules and put the reference on sys.path (idempotent)."""
    if "kod" in sys.modules:
        return
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
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def ref(module: str):
    install()
    return importlib.import_module(module)
