"""Oracle: YOLOv5 network (CSPDarknet + PAFPN + heads) in plain PyTorch fp32.

TEST INFRASTRUCTURE - see oracle/__init__.py.  Restates (does not import):

* conv->BN->SiLU unit          torchvision 0.15.2 ``Conv2dNormActivation`` as
                               used at kod/nn/layers/csp.py:30-46,81-89
* CSP block / CSP layer        kod/nn/layers/csp.py:16-111
* SPPF bottleneck              kod/nn/layers/sppf.py:14-84
* backbone stem + 4 stages     kod/nn/backbones/yolov5.py:19-132
* PAFPN neck                   kod/nn/necks/yolov5_pafpn.py:16-202
* box / obj / cls heads        kod/nn/heads/yolov5.py:12-178
* whole network                kod/nn/networks/yolov5.py:24-108
* channel rounding             kod/nn/utils.py:7-22

Module attribute names reproduce the reference's so ``state_dict()`` keys,
shapes and construction order (hence seeded initial weights) are identical.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import NamedTuple

import torch
import torch.nn as nn
import torch.nn.functional as F

BN_EPS, BN_MOMENTUM = 1e-3, 0.03            # kod/nn/networks/yolov5.py:24
# (in, out, blocks, identity, spp)            kod/nn/networks/yolov5.py:26-31
P5 = ((64, 128, 3, True, False), (128, 256, 6, True, False),
      (256, 512, 9, True, False), (512, 1024, 3, False, True))


def ch(x: float, widen: float, div: int = 8) -> int:       # kod/nn/utils.py:7-13
    return math.ceil(x * widen / div) * div


def depth(x: int, deepen: float) -> int:                     # kod/nn/utils.py:16-22
    return int(max(round(x * deepen), 1) if x > 1 else x)


class HeadOut(NamedTuple):                                   # kod/nn/heads/types.py:8-11
    box: torch.Tensor
    obj: torch.Tensor
    cls: torch.Tensor


class NetOut(NamedTuple):                                    # kod/nn/networks/yolov5.py:34-37
    ll: HeadOut
    ml: HeadOut
    hl: HeadOut


def cba(cin: int, cout: int, k: int = 1, s: int = 1, p: int | None = None) -> nn.Sequential:
    """conv(bias=False) -> BN(eps 1e-3, mom .03) -> SiLU, children named 0,1,2."""
    p = (k - 1) // 2 if p is None else p
    return nn.Sequential(nn.Conv2d(cin, cout, k, s, p, bias=False),
                         nn.BatchNorm2d(cout, eps=BN_EPS, momentum=BN_MOMENTUM),
                         nn.SiLU())


class Bottleneck(nn.Module):                                 # CSPBlock, csp.py:16-58
    def __init__(self, c: int, identity: bool):
        super().__init__()
        self.conv1 = cba(c, c, 1)
        self.conv2 = cba(c, c, 3, 1, 1)
        self.identity = identity

    def forward(self, x):
        y = self.conv2(self.conv1(x))
        return y + x if self.identity else y


class CSP(nn.Module):                                        # CSPLayer, csp.py:66-111
    def __init__(self, cin: int, cout: int, n: int, identity: bool):
        super().__init__()
        mid = int(cout * 0.5)
        self.short_conv = cba(cin, mid)
        self.main_conv = cba(cin, mid)
        self.last_conv = cba(2 * mid, cout)
        self.blocks = nn.Sequential(*[Bottleneck(mid, identity) for _ in range(n)])

    def forward(self, x):
        return self.last_conv(torch.cat([self.blocks(self.main_conv(x)), self.short_conv(x)], 1))


class SPPF(nn.Module):                                       # sppf.py:14-84
    def __init__(self, cin: int, cout: int, k=5, use_conv_first: bool = True, mid_channels_scale: float = 0.5):
        """k: int = the cascade cat[x, p(x), p(p(x)), p(p(p(x)))] (sppf.py:49-55,74-77); a sequence = parallel pools
        cat[x, p_k0(x), p_k1(x), ...] (sppf.py:56-63,78-82); use_conv_first=False: no conv1, mid = cin (sppf.py:37-39)"""
        super().__init__()
        mid = int(cin * mid_channels_scale) if use_conv_first else cin
        self.conv1 = cba(cin, mid) if use_conv_first else None
        self.k = k
        if isinstance(k, int):
            self.poolings = nn.MaxPool2d(k, 1, k // 2)
            n = 4
        else:
            self.poolings = nn.ModuleList([nn.MaxPool2d(q, 1, q // 2) for q in k])
            n = len(k) + 1
        self.conv2 = cba(n * mid, cout)

    def forward(self, x):
        if self.conv1 is not None:
            x = self.conv1(x)
        if isinstance(self.k, int):
            y1 = self.poolings(x)
            y2 = self.poolings(y1)
            return self.conv2(torch.cat([x, y1, y2, self.poolings(y2)], 1))
        return self.conv2(torch.cat([x] + [p(x) for p in self.poolings], 1))


class Stage(nn.Module):                                      # backbones/yolov5.py:27-82
    def __init__(self, cfg, widen: float, deepen: float):
        super().__init__()
        cin, cout = ch(cfg[0], widen), ch(cfg[1], widen)
        parts = [cba(cin, cout, 3, 2, 1), CSP(cout, cout, depth(cfg[2], deepen), cfg[3])]
        if cfg[4]:
            parts.append(SPPF(cout, cout))
        self.blocks = nn.Sequential(*parts)

    def forward(self, x):
        return self.blocks(x)


class Backbone(nn.Module):                                   # backbones/yolov5.py:85-132
    def __init__(self, widen: float, deepen: float):
        super().__init__()
        self.stem = cba(3, ch(P5[0][0], widen), 6, 2, 2)
        self.stages = nn.ModuleDict(OrderedDict(
            (f"stage{i + 1}", Stage(cfg, widen, deepen)) for i, cfg in enumerate(P5)))

    def forward(self, x):
        x = self.stem(x)
        outs = []
        for s in self.stages.values():
            x = s(x)
            outs.append(x)
        return outs


class Neck(nn.Module):                                       # necks/yolov5_pafpn.py:16-202
    def __init__(self, cs, widen: float, deepen: float, num_blocks: int = 3):
        super().__init__()
        c = [ch(v, widen) for v in cs]
        c2 = [ch(v * 2, widen) for v in cs]
        n = depth(num_blocks, deepen)
        last = len(cs) - 1
        self.reduce_layers = nn.ModuleList(
            [cba(c[i], c[i - 1]) if i == last else nn.Identity() for i in range(len(cs))])
        self.upsample_layers = nn.ModuleList()
        self.top_down_layers = nn.ModuleList()
        for i in range(last, 0, -1):
            self.upsample_layers.append(nn.Upsample(scale_factor=2, mode="nearest"))
            csp = CSP(c2[i - 1], c[i - 1], n, False)
            self.top_down_layers.append(csp if i == 1 else nn.Sequential(csp, cba(c[i - 1], c[i - 2])))
        self.downsample_layers = nn.ModuleList()
        self.bottom_up_layers = nn.ModuleList()
        for i in range(last):
            self.downsample_layers.append(cba(c[i], c[i], 3, 2, 1))
            self.bottom_up_layers.append(CSP(c2[i], c[i + 1], n, False))

    def forward(self, feats):
        red = [m(f) for m, f in zip(self.reduce_layers, feats)]
        inner = [red[-1]]
        last = len(feats) - 1
        for i in range(last, 0, -1):
            up = self.upsample_layers[last - i](inner[0])
            inner.insert(0, self.top_down_layers[last - i](torch.cat([up, red[i - 1]], 1)))
        outs = [inner[0]]
        for i in range(last):
            down = self.downsample_layers[i](outs[-1])
            outs.append(self.bottom_up_layers[i](torch.cat([down, inner[i + 1]], 1)))
        return tuple(outs)


class _SubHead(nn.Module):
    """One biased 1x1 conv + 'b (a p) h w -> b a h w p' view (heads/yolov5.py:12-136)."""

    def __init__(self, cin: int, na: int, p: int, bias_shift: float = 0.0):
        super().__init__()
        self.conv = nn.Conv2d(cin, na * p, 1)
        if bias_shift:
            with torch.no_grad():
                self.conv.bias.add_(bias_shift)
        self.na, self.p = na, p

    def forward(self, x):
        y = self.conv(x)
        b, _, h, w = y.shape
        return y.view(b, self.na, self.p, h, w).permute(0, 1, 3, 4, 2)


class Head(nn.Module):                                       # heads/yolov5.py:139-178
    def __init__(self, cin: int, na: int, nc: int, stride: int):
        super().__init__()
        self.box_head = _SubHead(cin, na, 4)
        self.obj_head = _SubHead(cin, na, 1, math.log(8 / (640 / stride) ** 2))       # :113-121
        self.cls_head = _SubHead(cin, na, nc, math.log(0.6 / (nc - 0.99999)))          # :65-73

    def forward(self, x):
        return HeadOut(self.box_head(x), self.obj_head(x), self.cls_head(x))


class OracleYolov5(nn.Module):                               # networks/yolov5.py:40-108
    def __init__(self, num_anchors_per_cell: int, num_classes: int,
                 widen_factor: float = 1.0, deepen_factor: float = 1.0):
        super().__init__()
        cs = [P5[1][1], P5[2][1], P5[3][1]]
        self.num_classes = num_classes
        self.backbone = Backbone(widen_factor, deepen_factor)
        self.neck = Neck(cs, widen_factor, deepen_factor)
        hc = [ch(v, widen_factor) for v in cs]
        self.ll_head = Head(hc[0], num_anchors_per_cell, num_classes, 8)
        self.ml_head = Head(hc[1], num_anchors_per_cell, num_classes, 16)
        self.hl_head = Head(hc[2], num_anchors_per_cell, num_classes, 32)

    def forward(self, x):
        _, p3, p4, p5 = self.backbone(x)
        p3, p4, p5 = self.neck([p3, p4, p5])
        return NetOut(self.ll_head(p3), self.ml_head(p4), self.hl_head(p5))
