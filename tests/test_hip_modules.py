"""The reference's building blocks as importable modules of their own (kod/nn/layers/csp.py, sppf.py,
backbones/yolov5.py, necks/yolov5_pafpn.py, heads/yolov5.py): same constructor / call signatures, state_dict keys and
seeded initial weights as the oracle's restatement (pinned to the reference), forward and backward on the HIP engine
against the fp32 oracle modules (bf16 storage tolerances; a handful of layers each, so no deep amplification)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import network as N  # noqa: E402
from object_detection_cib_amd.nn.backbones.yolov5 import StageConfig, Yolov5Backbone  # noqa: E402
from object_detection_cib_amd.nn.heads.yolov5 import (Yolov5Head, Yolov5BoxHead, Yolov5ObjectnessHead,  # noqa: E402
                                                       Yolov5ClassificationHead)
from object_detection_cib_amd.nn.layers.csp import CSPBlock, CSPLayer  # noqa: E402
from object_detection_cib_amd.nn.layers.sppf import SPPFBottleneck  # noqa: E402
from object_detection_cib_amd.nn.necks.yolov5_pafpn import Yolov5PAFPN  # noqa: E402
from object_detection_cib_amd.nn.networks.yolov5 import Yolov5BatchNorm2d  # noqa: E402


def _rel(a, b):
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _flat(o):
    if isinstance(o, torch.Tensor):
        return [o]
    out = []
    for x in o:
        out += _flat(x)
    return out


def _compare(hip, ref, xs, ftol=1e-2, gtol=4e-2, in_gtol=None, vs_emulation=False):
    """same seeded weights; forward outputs, input gradients and parameter gradients of sum(w_k * out_k).
    vs_emulation: deep stacks of train-mode BatchNorm amplify bf16 storage noise layer by layer, so the forward bar per
    output is what the fp32 oracle's own bf16-storage emulation (oracle/bf16_emul.py) shows against fp32 on this input:
    HIP must be no further from fp32 than 1.5 x the emulation + 5e-3."""
    import copy
    from oracle import bf16_emul
    emu_err = emu_gerr = None
    if vs_emulation:
        emu = bf16_emul.emulate(copy.deepcopy(ref).train(), image_too=len(xs) == 1 and xs[0].shape[1] == 3)
        fp32 = copy.deepcopy(ref).train()
        ge = torch.Generator().manual_seed(7)
        fp = _flat(fp32(xs[0] if len(xs) == 1 else xs))
        em = _flat(emu(xs[0] if len(xs) == 1 else xs))
        wse = [torch.randn(o.shape, generator=ge) for o in fp]
        emu_err = [_rel(a, b) for a, b in zip(em, fp)]
        sum((o * w).sum() for o, w in zip(fp, wse)).backward()
        sum((o * w).sum() for o, w in zip(em, wse)).backward()
        num = sum((p.grad.double() - q.grad.double()).pow(2).sum().item() for p, q in zip(emu.parameters(), fp32.parameters()))
        den = sum(q.grad.double().pow(2).sum().item() for q in fp32.parameters())
        emu_gerr = (num / den) ** 0.5
    assert list(hip.state_dict().keys()) == list(ref.state_dict().keys())
    for (k, a), b in zip(hip.state_dict().items(), ref.state_dict().values()):
        assert torch.equal(a, b), k
    hip = hip.cuda().train()
    ref = ref.train()
    g = torch.Generator().manual_seed(7)
    xr = [x.clone().requires_grad_(True) for x in xs]
    xh = [x.clone().cuda().requires_grad_(True) for x in xs]
    out_r = _flat(ref(xr[0]) if len(xr) == 1 else ref(xr))
    out_h = _flat(hip(xh[0]) if len(xh) == 1 else hip(xh))
    assert len(out_r) == len(out_h)
    ws = [torch.randn(o.shape, generator=g) for o in out_r]
    ftols = list(ftol) if isinstance(ftol, (list, tuple)) else [ftol] * len(out_r)
    if emu_err is not None:
        ftols = [1.5 * e + 5e-3 for e in emu_err]
    errs = [_rel(o_h, o_r) for o_h, o_r in zip(out_h, out_r)]
    for o_h, o_r, e, t in zip(out_h, out_r, errs, ftols):
        assert o_h.shape == o_r.shape
        assert e <= t, ("forward", errs, "bars", ftols)
    sum((o * w).sum() for o, w in zip(out_r, ws)).backward()
    sum((o * w.cuda()).sum() for o, w in zip(out_h, ws)).backward()
    for a, b in zip(xh, xr):
        # (the backbone's input is the image: its gradient is the one thing the engine never computes)
        if hip.graph.inputs and b.grad is not None and b.grad.abs().sum() > 0:
            assert a.grad is not None and _rel(a.grad, b.grad) <= (in_gtol or gtol), ("input grad", _rel(a.grad, b.grad))
    num = den = 0.0
    for (k, p), q in zip(hip.named_parameters(), ref.parameters()):
        assert p.grad is not None, k
        num += (p.grad.double().cpu() - q.grad.double()).pow(2).sum().item()
        den += q.grad.double().pow(2).sum().item()
    gbar = gtol if emu_gerr is None else 1.5 * emu_gerr + 2e-2      # (deep stacks: the emulation's own distance from fp32)
    assert (num / den) ** 0.5 <= gbar, ("parameter grads", (num / den) ** 0.5, "bar", gbar)
    # BatchNorm buffers moved like torch's
    sd_h, sd_r = hip.state_dict(), ref.state_dict()
    for k in sd_r:
        if k.endswith("running_var"):
            assert _rel(sd_h[k], sd_r[k]) <= 2e-2, k
        if k.endswith("num_batches_tracked"):
            assert int(sd_h[k]) == int(sd_r[k]) == 1


def test_csp_block_and_layer():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(4, 64, 24, 40, generator=g)
    torch.manual_seed(3); hip = CSPBlock(64, 64, 1.0, True, Yolov5BatchNorm2d)
    torch.manual_seed(3); ref = N.Bottleneck(64, True)
    _compare(hip, ref, [x])
    torch.manual_seed(4); hip = CSPLayer(64, 128, 0.5, True, 2, Yolov5BatchNorm2d, torch.nn.SiLU)
    torch.manual_seed(4); ref = N.CSP(64, 128, 2, True)
    _compare(hip, ref, [x])
    # norm_layer = any nn.BatchNorm2d: torch's defaults (eps 1e-5, momentum 0.1) against the oracle with its modules set the same
    torch.manual_seed(6); hip = CSPLayer(64, 64, norm_layer=torch.nn.BatchNorm2d)
    torch.manual_seed(6); ref = N.CSP(64, 64, 1, True)
    for m in ref.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eps, m.momentum = 1e-5, 0.1
    _compare(hip, ref, [x])
    rv_h = torch.cat([v.cpu().flatten() for k, v in hip.state_dict().items() if k.endswith("running_var")])
    rv_r = torch.cat([v.flatten() for k, v in ref.state_dict().items() if k.endswith("running_var")])
    assert _rel(rv_h, rv_r) <= 2e-2                                  # the momentum-0.1 update of both
    with pytest.raises(ValueError):
        CSPLayer(64, 64, norm_layer=lambda c: torch.nn.GroupNorm(4, c))
    with pytest.raises(ValueError):
        CSPLayer(64, 64, activation_layer=torch.nn.GELU)


def _swap_activation(module, make):
    """the oracle modules are built with SiLU: put another activation in every conv + BN + activation unit"""
    for parent in list(module.modules()):
        for name, child in list(parent.named_children()):
            if isinstance(child, torch.nn.SiLU):
                setattr(parent, name, make() if make is not None else torch.nn.Identity())
    return module


@pytest.mark.parametrize("act", ["relu", "leaky", "hardswish", "none"])
def test_layers_with_other_activations(act):
    """`activation_layer` other than SiLU (kod/nn/layers/csp.py:16-46, sppf.py:14-27 take any callable): ReLU, LeakyReLU(0.1),
    Hardswish and None (torchvision's Conv2dNormActivation: no activation) through the plain elementwise passes
    (csrc/bn_act.hip bn_act_*; the fused SiLU epilogues are off for such a module) - a CSP layer with two blocks and an SPPF
    block, forward, input gradient and parameter gradients against the oracle modules with the same activation."""
    make = {"relu": torch.nn.ReLU, "leaky": lambda: torch.nn.LeakyReLU(0.1), "hardswish": torch.nn.Hardswish, "none": None}[act]
    x = torch.randn(4, 64, 24, 40, generator=torch.Generator().manual_seed(31))
    torch.manual_seed(32); hip = CSPLayer(64, 128, 0.5, True, 2, Yolov5BatchNorm2d, make)
    torch.manual_seed(32); ref = _swap_activation(N.CSP(64, 128, 2, True), make)
    assert hip._act[0] != 0
    # activations with a kink route a gradient or not by the SIGN of z, which bf16 storage of the pre-activation flips for
    # the few per mille of elements next to zero: every flip is a whole gradient term (measured: ReLU input gradient 0.12
    # against the fp32 oracle through five units; the kernels themselves are held to torch in test_bn_act_passes_vs_torch)
    kink = 2.5e-1 if act != "none" else 6e-2
    _compare(hip, ref, [x], gtol=kink, in_gtol=kink)
    x2 = torch.randn(3, 128, 16, 16, generator=torch.Generator().manual_seed(33))
    torch.manual_seed(34); hip = SPPFBottleneck(128, 128, norm_layer=Yolov5BatchNorm2d, activation_layer=make)
    torch.manual_seed(34); ref = _swap_activation(N.SPPF(128, 128), make)
    _compare(hip, ref, [x2], ftol=2e-2, gtol=max(1.5e-1, kink), in_gtol=max(2e-1, kink))


def test_sppf_bottleneck():
    x = torch.randn(3, 128, 16, 16, generator=torch.Generator().manual_seed(2))
    torch.manual_seed(5); hip = SPPFBottleneck(128, 128, norm_layer=Yolov5BatchNorm2d)
    torch.manual_seed(5); ref = N.SPPF(128, 128)
    # bf16 input + two conv layers with K = 128 / 256 (forward measured 1.2e-2).  Gradients: activations rounded to bf16 tie far
    # more often than fp32 ones, so a 5x5 max-pool routes some gradients to another pixel than the fp32 oracle does (the pool
    # backward itself is pinned against torch on identical bf16 inputs in test_hip_ops): a wider bar on what passes the pools
    _compare(hip, ref, [x], ftol=2e-2, gtol=1.5e-1, in_gtol=2e-1)
    with pytest.raises(ValueError):
        SPPFBottleneck(64, 64, kernel_sizes=4)                     # even windows change the image size: refused like a bad config
    with pytest.raises(ValueError):
        SPPFBottleneck(64, 64, kernel_sizes=(3, 17))


@pytest.mark.parametrize("ks,first", [(3, True), ((3, 7), True), ((7,), False), (9, True), ((3, 5, 7), True)])
def test_sppf_other_windows(ks, first):
    """SPPFBottleneck with windows other than 5 (kod/nn/layers/sppf.py:27-67 takes any size or sequence): cascades of k x k
    pools, parallel pools that are no cascade ((3, 7): both read x and their gradients meet in x's slice), one window
    without the leading conv, and a sequence that IS a cascade ((3, 5, 7) = three 3 x 3 pools) - against the oracle module,
    whose forms 'k3', 'k3_7', 'k7_noconv' are pinned to the reference by tests/golden/sppf.npz."""
    cin = 128 if first else 64
    x = torch.randn(3, cin, 16, 16, generator=torch.Generator().manual_seed(21))
    torch.manual_seed(22); hip = SPPFBottleneck(cin, 128, kernel_sizes=ks, use_conv_first=first, norm_layer=Yolov5BatchNorm2d)
    torch.manual_seed(22); ref = N.SPPF(cin, 128, ks, first)
    _compare(hip, ref, [x], ftol=2e-2, gtol=1.5e-1, in_gtol=2e-1)


@pytest.mark.parametrize("first", [True, False])
def test_sppf_parallel_pools_and_no_leading_conv(first):
    """SPPFBottleneck's other forms (kod/nn/layers/sppf.py:37-39,56-63,78-82): the parallel pools of sizes (5, 9, 13) - the
    same arithmetic as the cascade the network uses - with and without the leading 1x1 conv, against the oracle module
    (pinned to the reference by tests/golden/sppf.npz)."""
    cin = 128 if first else 64
    x = torch.randn(3, cin, 16, 16, generator=torch.Generator().manual_seed(12))
    torch.manual_seed(15); hip = SPPFBottleneck(cin, 128, kernel_sizes=(5, 9, 13), use_conv_first=first, norm_layer=Yolov5BatchNorm2d)
    torch.manual_seed(15); ref = N.SPPF(cin, 128, (5, 9, 13), first)
    assert (hip.conv1 is None) == (not first)
    _compare(hip, ref, [x], ftol=2e-2, gtol=1.5e-1, in_gtol=2e-1)


def test_backbone_and_neck_and_head():
    widen, deepen = 0.25, 0.33
    x = torch.rand(4, 3, 256, 256, generator=torch.Generator().manual_seed(3))   # (deepest map 8 x 8 x 4 images: sane BN statistics)
    torch.manual_seed(6); hip = Yolov5Backbone(Yolov5BatchNorm2d, torch.nn.SiLU, [StageConfig(*s) for s in N.P5], deepen, widen)
    torch.manual_seed(6); ref = N.Backbone(widen, deepen)
    import copy
    feats = [f.detach() for f in copy.deepcopy(ref)(x)]            # (a copy: the forward moves the BatchNorm buffers)
    assert [tuple(f.shape[1:]) for f in feats] == [(32, 64, 64), (64, 32, 32), (128, 16, 16), (256, 8, 8)]
    # 7 / 13 / 22 / 33 conv + train-mode BatchNorm + SiLU layers deep: bf16 storage noise grows with depth against fp32
    # (measured HIP 1.1 / 2.0 / 3.3 / 9.9 e-2): held to the oracle's bf16-storage emulation, output by output
    _compare(hip, ref, [x], gtol=2e-1, vs_emulation=True)
    torch.manual_seed(7); hip = Yolov5PAFPN([256, 512, 1024], Yolov5BatchNorm2d, torch.nn.SiLU, 3, 0.5, deepen, widen)
    torch.manual_seed(7); ref = N.Neck([256, 512, 1024], widen, deepen)
    _compare(hip, ref, [f.clone() for f in feats[1:]], gtol=2e-1, vs_emulation=True)
    torch.manual_seed(8); hip = Yolov5Head(64, 3, 10, 8)
    torch.manual_seed(8); ref = N.Head(64, 3, 10, 8)
    _compare(hip, ref, [feats[1].clone()], ftol=5e-3, gtol=2e-2)


def test_head_pieces_as_modules():
    """Yolov5BoxHead / Yolov5ObjectnessHead / Yolov5ClassificationHead (kod/nn/heads/yolov5.py:12-136) as modules of their
    own: state_dict = {conv.weight, conv.bias} with the reference's seeded initialisation (bias shifts included), output
    [B, A, h, w, P], forward / input gradient / parameter gradients against the oracle's per-piece head."""
    import math
    x = torch.randn(3, 64, 20, 12, generator=torch.Generator().manual_seed(21))
    for seed, hip_fn, ref_fn, P in (
            (31, lambda: Yolov5BoxHead(64, 3), lambda: N._SubHead(64, 3, 4), 4),
            (32, lambda: Yolov5ObjectnessHead(64, 3, 16), lambda: N._SubHead(64, 3, 1, math.log(8 / (640 / 16) ** 2)), 1),
            (33, lambda: Yolov5ObjectnessHead(64, 3, 16, 0.02, False), lambda: N._SubHead(64, 3, 1, -math.log(0.98 / 0.02)), 1),
            (34, lambda: Yolov5ClassificationHead(64, 3, 10), lambda: N._SubHead(64, 3, 10, math.log(0.6 / (10 - 0.99999))), 10)):
        torch.manual_seed(seed); hip = hip_fn()
        torch.manual_seed(seed); ref = ref_fn()
        assert list(hip.state_dict().keys()) == ["conv.weight", "conv.bias"]
        out = hip.cuda()(x.cuda())
        assert tuple(out.shape) == (3, 3, 20, 12, P)
        torch.manual_seed(seed); hip = hip_fn()
        _compare(hip, ref, [x.clone()], ftol=5e-3, gtol=2e-2)
