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

from dataclasses import dataclass, field
from typing import List, Optional

# (in, out, blocks, identity, spp)   kod/nn/networks/yolov5.py:26-31
P5_STAGES = ((64, 128, 3, True, False), (128, 256, 6, True, False),
             (256, 512, 9, True, False), (512, 1024, 3, False, True))


from ..nn.utils import make_divisible, make_round  # noqa: E402,F401  (channel / depth rounding; also re-exported from here)


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


def head_param(h: "HeadUnit", key: str, what: str) -> str:
    """state_dict path of a head conv parameter: '<head>.<box|obj|cls>_head.conv.<weight|bias>' (heads/yolov5.py:139-178);
    a Yolov5Head that is a module of its own has no '<head>.' prefix."""
    return (h.name + "." if h.name else "") + f"{key}_head.conv.{what}"


@dataclass
class Op:
    kind: str            # conv | pool | up | head
    unit: object = None
    src: Optional[View] = None
    dst: Optional[View] = None
    k: int = 5           # pool: window (K x K / stride 1 / pad K // 2)


@dataclass
class Graph:
    num_anchors: int
    num_classes: int
    units: List[ConvUnit] = field(default_factory=list)      # construction (= reference RNG) order
    heads: List[HeadUnit] = field(default_factory=list)
    ops: List[Op] = field(default_factory=list)              # execution order
    bufs: List[Buf] = field(default_factory=list)
    # sub-network graphs (nn/layers, nn/backbones, nn/necks, nn/heads as modules of their own): activation views the
    # caller fills / reads.  The whole network has neither: its input is the image buffer, its outputs are the heads.
    inputs: List[View] = field(default_factory=list)
    outputs: List[View] = field(default_factory=list)


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

    def csp(self, name, src: View, dst: View, cin, cout, n, identity, expand_ratio: float = 0.5):
        """CSPLayer (csp.py:66-111): registration order short, main, last, blocks; exec order main, short, blocks,
        last (main and short read the same input and are independent: back to back, their SyncBN statistic
        exchanges travel as one grouped collective).  cat([main_branch, short]) is the buffer `cat`.
        name = "" for a CSPLayer that is a module of its own (parameter paths without a prefix)."""
        stride = src.stride
        mid = int(cout * expand_ratio)
        name = name or "_"
        pre = "" if name == "_" else name + "."
        short = self.unit(f"{pre}short_conv", cin, mid)
        main = self.unit(f"{pre}main_conv", cin, mid)
        last = self.unit(f"{pre}last_conv", 2 * mid, cout)
        blocks = [(self.unit(f"{pre}blocks.{j}.conv1", mid, mid),
                   self.unit(f"{pre}blocks.{j}.conv2", mid, mid, 3, 1, 1)) for j in range(n)]
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

    def sppf(self, name, src: Optional[View], dst: View, cin, cout, mid_channels_scale: float = 0.5, use_conv_first: bool = True,
             kernel_sizes=5):
        """SPPFBottleneck (sppf.py:14-84): conv2(cat[x, pools...]), x = conv1(in).  kernel_sizes: an int k = the cascade
        cat[x, p(x), p(p(x)), p(p(p(x)))] of k x k pools (sppf.py:49-55,74-77); a sequence = parallel pools cat[x, p_k0(x),
        p_k1(x), ...] (sppf.py:56-63,78-82) - run as a cascade of the first size where the sequence is one (k, 2k - 1,
        3k - 2, ...: a stride-1 max-pool of k applied j times IS the max-pool of j (k - 1) + 1, values and gradient routing
        alike; the SPP sequence (5, 9, 13)), else each pool reads x.  use_conv_first=False (sppf.py:37-39): no conv1, the
        module's input IS the first slice of the concat buffer (src = None: the caller takes the returned view as its input)."""
        pre = name + "." if name else ""
        name = name or "_"
        mid = int(cin * mid_channels_scale) if use_conv_first else cin
        plan = sppf_pool_plan(kernel_sizes)                 # [(source slice, window)] per pool; slice q + 1 is its output
        n = len(plan) + 1
        s1 = self.unit(f"{pre}conv1", cin, mid) if use_conv_first else None
        s2 = self.unit(f"{pre}conv2", n * mid, cout)
        stride = src.stride if src is not None else dst.stride
        scat = self.buf(f"{name}.cat", stride, n * mid)
        if s1 is not None:
            self.run(s1, src, View(scat, 0, mid))
        for q, (sq, k) in enumerate(plan):
            self.g.ops.append(Op("pool", None, View(scat, sq * mid, mid), View(scat, (q + 1) * mid, mid), k))
        self.run(s2, self.full(scat), dst)
        return View(scat, 0, mid)


def _backbone(b: _Builder, g: Graph, pre: str, stages, widen_factor: float, deepen_factor: float, stage_dst: dict,
              expand_ratio: float = 0.5, spp_kernel_sizes=5):
    """Yolov5Backbone (backbones/yolov5.py:85-132): 6x6/s2/p2 stem + stages (3x3/s2 conv, CSPLayer, SPPF on `use_spp`).
    stage_dst: {stage index (1-based): View the stage writes its output into} (else a buffer of its own).
    Returns the stage outputs."""
    md = lambda v: make_divisible(v, widen_factor)
    image = b.buf("image", 1, 8)                       # pixel pairs x 4 channels, see misc_ops.hip
    g.image = image
    c0 = md(stages[0][0])
    stem = b.unit(f"{pre}stem", 8, c0, 6, 2, 2, stem=True)
    cur = b.full(b.buf(f"{pre}stem.out", 2, c0))
    b.run(stem, b.full(image), cur)
    outs = []
    for i, (ci, co, nb, ident, spp) in enumerate(stages, start=1):
        cin, cout = md(ci), md(co)
        sname = f"{pre}stages.stage{i}.blocks"
        stride = 2 ** (i + 1)
        conv = b.unit(f"{sname}.0", cin, cout, 3, 2, 1)
        x = b.full(b.buf(f"{sname}.0.out", stride, cout))
        b.run(conv, cur, x)
        dst = stage_dst.get(i) or b.full(b.buf(f"{sname}.1.out", stride, cout))
        b.csp(f"{sname}.1", x, dst, cout, cout, make_round(nb, deepen_factor), ident, expand_ratio)
        cur = dst
        if spp:                                        # SPPFBottleneck (sppf.py:14-84; backbones/yolov5.py:68-76 spp_kernel_sizes)
            P5 = b.full(b.buf(f"{sname}.2.out", stride, cout))
            b.sppf(f"{sname}.2", cur, P5, cout, cout, kernel_sizes=spp_kernel_sizes)
            cur = P5
        outs.append(cur)
    return outs


def _neck(b: _Builder, g: Graph, pre: str, cs, P5: View, td0cat: Buf, td1cat: Buf, n_neck: int, expand_ratio: float = 0.5):
    """Yolov5PAFPN (necks/yolov5_pafpn.py:16-202) over P3 = td1cat[cs0:], P4 = td0cat[cs1:], P5.  Returns (T3, O4, O5)."""
    bu0cat = b.buf(f"{pre}bu0.cat", 16, 2 * cs[0])      # [down(T3) | T4]
    bu1cat = b.buf(f"{pre}bu1.cat", 32, 2 * cs[1])      # [down(O4) | R5]
    red = b.unit(f"{pre}reduce_layers.2", cs[2], cs[1])
    R5 = View(bu1cat, cs[1], cs[1])
    b.run(red, P5, R5)
    # top-down idx=2: Sequential(CSPLayer, 1x1 reduce) ; idx=1: CSPLayer
    g.ops.append(Op("up", None, R5, View(td0cat, 0, cs[1])))
    t4p = b.full(b.buf(f"{pre}td0.csp.out", 16, cs[1]))
    b.csp(f"{pre}top_down_layers.0.0", b.full(td0cat), t4p, 2 * cs[1], cs[1], n_neck, False, expand_ratio)
    tdr = b.unit(f"{pre}top_down_layers.0.1", cs[1], cs[0])
    T4 = View(bu0cat, cs[0], cs[0])
    b.run(tdr, t4p, T4)
    g.ops.append(Op("up", None, T4, View(td1cat, 0, cs[0])))
    T3 = b.full(b.buf(f"{pre}out.ll", 8, cs[0]))
    b.csp(f"{pre}top_down_layers.1", b.full(td1cat), T3, 2 * cs[0], cs[0], n_neck, False, expand_ratio)
    # bottom-up
    d0 = b.unit(f"{pre}downsample_layers.0", cs[0], cs[0], 3, 2, 1)
    b.run(d0, T3, View(bu0cat, 0, cs[0]))
    O4 = b.full(b.buf(f"{pre}out.ml", 16, cs[1]))
    b.csp(f"{pre}bottom_up_layers.0", b.full(bu0cat), O4, 2 * cs[0], cs[1], n_neck, False, expand_ratio)
    d1 = b.unit(f"{pre}downsample_layers.1", cs[1], cs[1], 3, 2, 1)
    b.run(d1, O4, View(bu1cat, 0, cs[1]))
    O5 = b.full(b.buf(f"{pre}out.hl", 32, cs[2]))
    b.csp(f"{pre}bottom_up_layers.1", b.full(bu1cat), O5, 2 * cs[1], cs[2], n_neck, False, expand_ratio)
    return T3, O4, O5


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
    P3 = View(td1cat, cs[0], cs[0])
    P4 = View(td0cat, cs[1], cs[1])
    outs = _backbone(b, g, "backbone.", P5_STAGES, widen_factor, deepen_factor, {2: P3, 3: P4})
    T3, O4, O5 = _neck(b, g, "neck.", cs, outs[3], td0cat, td1cat, n_neck)
    # ---- heads (networks/yolov5.py:86-88)
    for name, src, stride in (("ll_head", T3, 8), ("ml_head", O4, 16), ("hl_head", O5, 32)):
        h = HeadUnit(name, src.C, stride, src)
        g.heads.append(h)
        g.ops.append(Op("head", h, src, None))
    return g


# ---- sub-networks as graphs of their own (the reference's nn.Module classes of the same names) -----------------------
def build_csp_block_graph(cin: int, cout: int, expand_ratio: float = 0.5, add_identity: bool = True) -> Graph:
    """CSPBlock (csp.py:16-58): conv2_3x3(conv1_1x1(x)) [+ x]."""
    g = Graph(0, 0)
    b = _Builder(g)
    hidden = int(cout * expand_ratio)
    c1 = b.unit("conv1", cin, hidden)
    c2 = b.unit("conv2", hidden, cout, 3, 1, 1)
    x = b.full(b.buf("in", 1, cin))
    h = b.full(b.buf("hid", 1, hidden))
    y = b.full(b.buf("out", 1, cout))
    b.run(c1, x, h)
    b.run(c2, h, y, residual=x if (add_identity and cin == cout) else None)
    g.inputs, g.outputs = [x], [y]
    return g


def build_csp_layer_graph(cin: int, cout: int, expand_ratio: float = 0.5, add_identity: bool = True, num_blocks: int = 1) -> Graph:
    """CSPLayer (csp.py:66-111)."""
    g = Graph(0, 0)
    b = _Builder(g)
    x = b.full(b.buf("in", 1, cin))
    y = b.full(b.buf("out", 1, cout))
    b.csp("", x, y, cin, cout, num_blocks, add_identity, expand_ratio)
    g.inputs, g.outputs = [x], [y]
    return g


def sppf_pool_plan(kernel_sizes):
    """[(index of the concat slice the pool reads, window)] for SPPFBottleneck's kernel_sizes (sppf.py:27-83)."""
    def ok(k):
        if not (isinstance(k, int) and 1 <= k <= 15 and k % 2 == 1):
            raise ValueError(f"SPPF window {k!r}: odd sizes 1 .. 15 (stride-1 pools that keep the image size)")
        return k
    if isinstance(kernel_sizes, int):
        return [(q, ok(kernel_sizes)) for q in range(3)]                     # the cascade
    ks = [ok(int(k)) for k in kernel_sizes]
    if not ks:
        raise ValueError("SPPF needs at least one window")
    if ks[0] > 1 and all(k == (j + 1) * (ks[0] - 1) + 1 for j, k in enumerate(ks)):
        return [(q, ks[0]) for q in range(len(ks))]                          # parallel pools that ARE a cascade of the first
    return [(0, k) for k in ks]                                              # parallel pools of x


def build_sppf_graph(cin: int, cout: int, mid_channels_scale: float = 0.5, use_conv_first: bool = True, kernel_sizes=5) -> Graph:
    """SPPFBottleneck (sppf.py:14-84) in every form: one window size in cascade or a sequence of parallel pools (see
    _Builder.sppf), with or without the leading 1x1 conv."""
    g = Graph(0, 0)
    b = _Builder(g)
    y = b.full(b.buf("out", 1, cout))
    if use_conv_first:
        x = b.full(b.buf("in", 1, cin))
        b.sppf("", x, y, cin, cout, mid_channels_scale, kernel_sizes=kernel_sizes)
    else:
        x = b.sppf("", None, y, cin, cout, mid_channels_scale, use_conv_first=False, kernel_sizes=kernel_sizes)
    g.inputs, g.outputs = [x], [y]
    return g


def build_backbone_graph(stages, widen_factor: float = 1.0, deepen_factor: float = 1.0, spp_kernel_sizes=5) -> Graph:
    """Yolov5Backbone (backbones/yolov5.py:85-132): image -> the four stage outputs."""
    g = Graph(0, 0)
    b = _Builder(g)
    g.outputs = _backbone(b, g, "", [tuple(s) for s in stages], widen_factor, deepen_factor, {}, spp_kernel_sizes=spp_kernel_sizes)
    return g


def build_pafpn_graph(in_channels_list, num_blocks: int = 3, expand_ratio: float = 0.5, deepen_factor: float = 1.0,
                      widen_factor: float = 1.0) -> Graph:
    """Yolov5PAFPN (necks/yolov5_pafpn.py:16-202): (P3, P4, P5) -> (T3, O4, O5); strides 8 / 16 / 32 of a virtual image."""
    g = Graph(0, 0)
    b = _Builder(g)
    cs = [make_divisible(c, widen_factor) for c in in_channels_list]
    assert len(cs) == 3, "the HIP neck is built for three pyramid levels"
    td0cat = b.buf("td0.cat", 16, 2 * cs[1])
    td1cat = b.buf("td1.cat", 8, 2 * cs[0])
    P5 = b.full(b.buf("in.p5", 32, cs[2]))
    g.inputs = [View(td1cat, cs[0], cs[0]), View(td0cat, cs[1], cs[1]), P5]
    g.outputs = list(_neck(b, g, "", cs, P5, td0cat, td1cat, make_round(num_blocks, deepen_factor), expand_ratio))
    return g


def build_head_graph(cin: int, num_anchors_per_cell: int, num_classes: int, stride: int) -> Graph:
    """Yolov5Head (heads/yolov5.py:139-178): the three biased 1x1 convs of one level as one GEMM."""
    g = Graph(num_anchors_per_cell, num_classes)
    b = _Builder(g)
    x = b.full(b.buf("in", stride, cin))
    h = HeadUnit("", cin, stride, x)
    g.heads.append(h)
    g.ops.append(Op("head", h, x, None))
    g.inputs = [x]
    return g
