"""Static program of the YOLOv5 network for the HIP executor.

The reference builds the network as nested nn.Modules and lets autograd trace it
(kod/nn/networks/yolov5.py:40-108, backbones/yolov5.py:85-132, necks/yolov5_pafpn.py:16-202,
layers/csp.py:16-111, layers/sppf.py:14-84, heads/yolov5.py:139-178).  Shapes never change during
training, so here the same topology is emitted ONCE as a flat list of kernel-level ops over named
channels-last buffers; torch.cat / nn.Upsample become channel-slice writes into pre-allocated concat
buffers.  Parameter names, shapes and construction order equal the reference's so state_dict keys and
seeded initial weights are identical.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional

# (in, out, blocks, identity, spp)   kod/nn/networks/yolov5.py:26-31
P5_STAGES = ((64, 128, 3, True, False), (128, 256, 6, True, False),
             (256, 512, 9, True, False), (512, 1024, 3, False, True))


def make_divisible(x: float, widen_factor: float = 1.0, divisor: int = 8) -> int:
    """kod/nn/utils.py:7-13."""
    return math.ceil(x * widen_factor / divisor) * divisor


def make_round(x: float, deepen_factor: float = 1.0) -> int:
    """kod/nn/utils.py:16-22."""
    return int(max(round(x * deepen_factor), 1) if x > 1 else x)


@dataclass
class Buf:
    name: str
    stride: int          # spatial stride w.r.t. the input image (H = img_h // stride)
    C: int


@dataclass
class View:
    buf: Buf
    coff: int
    C: int

    @property
    def stride(self):
        return self.buf.stride


@dataclass
class ConvUnit:
    """conv(bias=False) -> BN -> SiLU (torchvision Conv2dNormActivation); param path = name + '.0' / '.1'."""
    name: str
    cin: int
    cout: int
    k: int
    s: int
    p: int
    src: Optional[View] = None
    dst: Optional[View] = None
    residual: Optional[View] = None
    stem: bool = False
    sibling: Optional["ConvUnit"] = None      # CSP main_conv -> its short_conv: same input, executed back to back


@dataclass
class HeadUnit:
    name: str            # ll_head / ml_head / hl_head
    cin: int
    stride: int
    src: Optional[View] = None


@dataclass
class Op:
    kind: str            # conv | pool | up | head
    unit: object = None
    src: Optional[View] = None
    dst: Optional[View] = None


@dataclass
class Graph:
    num_anchors: int
    num_classes: int
    units: List[ConvUnit] = field(default_factory=list)      # construction (= reference RNG) order
    heads: List[HeadUnit] = field(default_factory=list)
    ops: List[Op] = field(default_factory=list)              # execution order
    bufs: List[Buf] = field(default_factory=list)


class _Builder:
    def __init__(self, g: Graph):
        self.g = g

    def buf(self, name, stride, C) -> Buf:
        b = Buf(name, stride, C)
        self.g.bufs.append(b)
        return b

    def full(self, b: Buf) -> View:
        return View(b, 0, b.C)

    def unit(self, name, cin, cout, k=1, s=1, p=None, stem=False) -> ConvUnit:
        u = ConvUnit(name, cin, cout, k, s, (k - 1) // 2 if p is None else p, stem=stem)
        self.g.units.append(u)
        return u

    def run(self, u: ConvUnit, src: View, dst: View, residual: Optional[View] = None):
        assert src.C == u.cin and dst.C == u.cout, (u.name, src.C, u.cin, dst.C, u.cout)
        u.src, u.dst, u.residual = src, dst, residual
        self.g.ops.append(Op("conv", u, src, dst))

    def csp(self, name, src: View, dst: View, cin, cout, n, identity):
        """CSPLayer (csp.py:66-111): registration order short, main, last, blocks; exec order main, short, blocks,
        last (main and short read the same input and are independent: back to back, their SyncBN statistic
        exchanges travel as one grouped collective).  cat([main_branch, short]) is the buffer `cat`."""
        stride = src.stride
        mid = int(cout * 0.5)
        short = self.unit(f"{name}.short_conv", cin, mid)
        main = self.unit(f"{name}.main_conv", cin, mid)
        last = self.unit(f"{name}.last_conv", 2 * mid, cout)
        blocks = [(self.unit(f"{name}.blocks.{j}.conv1", mid, mid),
                   self.unit(f"{name}.blocks.{j}.conv2", mid, mid, 3, 1, 1)) for j in range(n)]
        cat = self.buf(f"{name}.cat", stride, 2 * mid)
        cur = self.full(self.buf(f"{name}.m0", stride, mid))
        self.run(main, src, cur)
        self.run(short, src, View(cat, mid, mid))
        main.sibling = short
        for j, (c1, c2) in enumerate(blocks):
            hid = self.full(self.buf(f"{name}.b{j}.h", stride, mid))
            self.run(c1, cur, hid)
            out = View(cat, 0, mid) if j == n - 1 else self.full(self.buf(f"{name}.m{j + 1}", stride, mid))
            self.run(c2, hid, out, residual=cur if identity else None)
            cur = out
        self.run(last, self.full(cat), dst)


def build_graph(num_anchors_per_cell: int, num_classes: int, widen_factor: float = 1.0,
                deepen_factor: float = 1.0) -> Graph:
    g = Graph(num_anchors_per_cell, num_classes)
    b = _Builder(g)
    md = lambda v: make_divisible(v, widen_factor)
    cs = [md(P5_STAGES[1][1]), md(P5_STAGES[2][1]), md(P5_STAGES[3][1])]     # P3, P4, P5 channels
    n_neck = make_round(3, deepen_factor)

    # concat buffers of the neck (yolov5_pafpn.py:186,199); backbone outputs are written straight into them
    td0cat = b.buf("neck.td0.cat", 16, 2 * cs[1])      # [up(R5) | P4]
    td1cat = b.buf("neck.td1.cat", 8, 2 * cs[0])       # [up(T4) | P3]
    bu0cat = b.buf("neck.bu0.cat", 16, 2 * cs[0])      # [down(T3) | T4]
    bu1cat = b.buf("neck.bu1.cat", 32, 2 * cs[1])      # [down(O4) | R5]
    P3 = View(td1cat, cs[0], cs[0])
    P4 = View(td0cat, cs[1], cs[1])

    # ---- backbone (backbones/yolov5.py:85-132)
    image = b.buf("image", 1, 8)                       # pixel pairs x 4 channels, see misc_ops.hip
    g.image = image
    c0 = md(P5_STAGES[0][0])
    stem = b.unit("backbone.stem", 8, c0, 6, 2, 2, stem=True)
    cur = b.full(b.buf("backbone.stem.out", 2, c0))
    b.run(stem, b.full(image), cur)
    stage_dst = {1: None, 2: P3, 3: P4, 4: None}
    for i, (ci, co, nb, ident, spp) in enumerate(P5_STAGES, start=1):
        cin, cout = md(ci), md(co)
        sname = f"backbone.stages.stage{i}.blocks"
        stride = 2 ** (i + 1)
        conv = b.unit(f"{sname}.0", cin, cout, 3, 2, 1)
        x = b.full(b.buf(f"{sname}.0.out", stride, cout))
        b.run(conv, cur, x)
        dst = stage_dst[i] or b.full(b.buf(f"{sname}.1.out", stride, cout))
        b.csp(f"{sname}.1", x, dst, cout, cout, make_round(nb, deepen_factor), ident)
        cur = dst
        if spp:                                        # SPPFBottleneck (sppf.py:14-84), kernel 5
            mid = int(cout * 0.5)
            s1 = b.unit(f"{sname}.2.conv1", cout, mid)
            s2 = b.unit(f"{sname}.2.conv2", 4 * mid, cout)
            scat = b.buf(f"{sname}.2.cat", stride, 4 * mid)
            b.run(s1, cur, View(scat, 0, mid))
            for q in range(3):
                g.ops.append(Op("pool", None, View(scat, q * mid, mid), View(scat, (q + 1) * mid, mid)))
            P5 = b.full(b.buf(f"{sname}.2.out", stride, cout))
            b.run(s2, b.full(scat), P5)
            cur = P5

    # ---- neck (necks/yolov5_pafpn.py:16-202)
    red = b.unit("neck.reduce_layers.2", cs[2], cs[1])
    R5 = View(bu1cat, cs[1], cs[1])
    b.run(red, cur, R5)
    # top-down idx=2: Sequential(CSPLayer, 1x1 reduce) ; idx=1: CSPLayer
    g.ops.append(Op("up", None, R5, View(td0cat, 0, cs[1])))
    t4p = b.full(b.buf("neck.td0.csp.out", 16, cs[1]))
    b.csp("neck.top_down_layers.0.0", b.full(td0cat), t4p, 2 * cs[1], cs[1], n_neck, False)
    tdr = b.unit("neck.top_down_layers.0.1", cs[1], cs[0])
    T4 = View(bu0cat, cs[0], cs[0])
    b.run(tdr, t4p, T4)
    g.ops.append(Op("up", None, T4, View(td1cat, 0, cs[0])))
    T3 = b.full(b.buf("neck.out.ll", 8, cs[0]))
    b.csp("neck.top_down_layers.1", b.full(td1cat), T3, 2 * cs[0], cs[0], n_neck, False)
    # bottom-up
    d0 = b.unit("neck.downsample_layers.0", cs[0], cs[0], 3, 2, 1)
    b.run(d0, T3, View(bu0cat, 0, cs[0]))
    O4 = b.full(b.buf("neck.out.ml", 16, cs[1]))
    b.csp("neck.bottom_up_layers.0", b.full(bu0cat), O4, 2 * cs[0], cs[1], n_neck, False)
    d1 = b.unit("neck.downsample_layers.1", cs[1], cs[1], 3, 2, 1)
    b.run(d1, O4, View(bu1cat, 0, cs[1]))
    O5 = b.full(b.buf("neck.out.hl", 32, cs[2]))
    b.csp("neck.bottom_up_layers.1", b.full(bu1cat), O5, 2 * cs[1], cs[2], n_neck, False)

    # ---- heads (networks/yolov5.py:86-88)
    for name, src, stride in (("ll_head", T3, 8), ("ml_head", O4, 16), ("hl_head", O5, 32)):
        h = HeadUnit(name, src.C, stride, src)
        g.heads.append(h)
        g.ops.append(Op("head", h, src, None))
    return g
