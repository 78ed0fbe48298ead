"""GPU parity tests of the individual HIP kernels against plain torch fp32 on the same (bf16-rounded)
inputs.  All calls go through the C ABI (ctypes).  Tolerances: the kernels accumulate in fp32 and round
outputs to bf16 (rel 2^-9), so comparisons use rtol 1e-2 on bf16 outputs and 2e-3 on fp32 outputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from object_detection_cib_amd import _lib  # noqa: E402
from hip_helpers import bf, nchw, nhwc, pack, pad, stream, conv_fwd_raw  # noqa: E402


def _close(a, b, rtol, atol, what=""):
    a, b = a.double(), b.double()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = (err > tol).sum().item()
    assert bad == 0, f"{what}: {bad}/{a.numel()} off, max err {err.max().item():.4g}, ref max {b.abs().max().item():.4g}"


CONV_CASES = [
    # B, Cin, H, W, Cout, k, s, p
    (2, 32, 16, 16, 32, 1, 1, 0),
    (2, 64, 20, 12, 64, 1, 1, 0),
    (1, 128, 8, 8, 256, 1, 1, 0),
    (2, 32, 16, 16, 32, 3, 1, 1),
    (2, 64, 12, 20, 128, 3, 1, 1),
    (2, 32, 16, 16, 64, 3, 2, 1),
    (1, 128, 10, 10, 128, 3, 2, 1),
    (3, 48, 9, 7, 96, 3, 1, 1),          # yv5m-like channel counts, odd spatial dims
    (1, 512, 4, 4, 512, 1, 1, 0),
    (5, 64, 60, 56, 160, 3, 1, 1),       # M = 16800, K = 576: the 256-pixel-tile / 3-stage-ring configuration, ragged M and N
    (4, 128, 64, 64, 128, 3, 2, 1),      # same configuration through the stride-2 forward and its parity-class dgrad
    (2, 48, 12, 10, 48, 1, 1, 0),        # Cin % 16 == 0 only: half-step FAST form with a K tail (48 -> 64), both directions
    (2, 16, 8, 8, 48, 3, 1, 1),          # 16 input channels: every half step is a new tap
    (2, 96, 10, 10, 48, 3, 2, 1),        # stride-2 dgrad gathers 48-channel dY: half-step form in the merged launch
    (3, 80, 7, 9, 80, 3, 1, 1),          # yv5x-like 80 channels
    # the CSP blocks' 3x3 layers at their real widths (weight gradient: conv_wgrad_row3_kernel - image rows of 160 / 80 /
    # 40 / 20 pixels against 32-row reduction steps, border masks, every wave layout)
    (2, 32, 160, 160, 32, 3, 1, 1),
    (2, 64, 80, 80, 64, 3, 1, 1),
    (3, 128, 40, 40, 128, 3, 1, 1),
    (5, 256, 20, 20, 256, 3, 1, 1),
    (2, 64, 24, 40, 32, 3, 1, 1),        # N = 32 with two channel chunks
    (2, 32, 10, 6, 64, 3, 1, 1),         # image rows shorter than a DMA piece
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(case):
    B, Cin, H, W, Cout, k, s, p = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = bf(torch.randn(B, Cin, H, W, generator=g))
    w = bf(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y = F.conv2d(xr, wr, None, s, p)
    dy = bf(torch.randn(y.shape, generator=g))
    y.backward(dy)
    lib = _lib.lib()
    pk = pack([w])
    xb = nhwc(x)
    yb, stats = conv_fwd_raw(xb, (0, Cin), pk, s, p)
    got = nchw(yb)
    _close(got, y.detach(), 1e-2, 2e-2, "fwd")
    # BN partial statistics are sums over the stored (bf16-rounded) outputs
    ssum = stats[0].sum(-1).cpu().double()
    ssq = stats[1].sum(-1).cpu().double()
    _close(ssum, got.double().sum((0, 2, 3)), 1e-4, 1e-2, "stats sum")
    _close(ssq, (got.double() ** 2).sum((0, 2, 3)), 1e-4, 1e-2, "stats sumsq")
    # dgrad
    Ho, Wo = y.shape[2:]
    dyb = nhwc(dy)
    dxb = torch.zeros((B, H, W, Cin), dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.kodhip_conv_dgrad(dyb.data_ptr(), pk["d"].data_ptr(), dxb.data_ptr(), B, H, W, Cin, 0, Cin,
                                     Cout, k, k, s, s, p, p, pk["Kdp"], Cout, 0, 0, None, stream()), "dgrad")
    _close(nchw(dxb), xr.grad, 1e-2, 3e-2, "dgrad")
    # accumulate form
    _lib.check(lib.kodhip_conv_dgrad(dyb.data_ptr(), pk["d"].data_ptr(), dxb.data_ptr(), B, H, W, Cin, 0, Cin,
                                     Cout, k, k, s, s, p, p, pk["Kdp"], Cout, 0, 1, None, stream()), "dgrad acc")
    _close(nchw(dxb), 2 * xr.grad, 2e-2, 6e-2, "dgrad accumulate")
    # wgrad
    M = B * Ho * Wo
    splits = lib.kodhip_conv_wgrad_splits_geo(B, H, W, Cin, Cin, Cout, k, k, s, s, p, p, pk["Kp"], Cout)
    part = torch.zeros(splits * Cout * pk["Kp"], dtype=torch.float32, device="cuda")
    gw = torch.zeros_like(w, device="cuda")
    _lib.check(lib.kodhip_conv_wgrad(xb.data_ptr(), dyb.data_ptr(), part.data_ptr(), gw.data_ptr(), B, H, W, Cin, 0,
                                     Cin, Cout, k, k, s, s, p, p, pk["Kp"], Cout, 0, Cout, 0, 1.0, stream()), "wgrad")
    _close(gw.cpu(), wr.grad, 2e-3, 2e-3 * wr.grad.abs().max().item(), "wgrad")


@pytest.mark.parametrize("form", ["classes", "folded"])
@pytest.mark.parametrize("case", [(2, 32, 16, 16, 64), (1, 128, 10, 12, 128), (3, 48, 6, 8, 96), (1, 256, 4, 4, 512),
                                  (4, 32, 96, 64, 64)])
def test_conv_dgrad_stride2_parity_classes(case, form):
    """3x3/s2/p1 data gradient through the 4 parity-class problems of one launch (kodhip_conv_dgrad_s2) and through the
    folded form (kodhip_conv_dgrad_s2f: one 2x2-tap gather, 4 x Cin columns, depth-to-space epilogue)
    (+ accumulate form, channel slice)."""
    B, Cin, H, W, Cout = case
    g = torch.Generator().manual_seed(sum(case))
    x = bf(torch.randn(B, Cin, H, W, generator=g)).requires_grad_(True)
    w = bf(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)
    y = F.conv2d(x, w, None, 2, 1)
    dy = bf(torch.randn(y.shape, generator=g))
    y.backward(dy)
    lib = _lib.lib()
    pk = pack([w], s2=True if form == "classes" else "fold")
    fn = lib.kodhip_conv_dgrad_s2 if form == "classes" else lib.kodhip_conv_dgrad_s2f
    dyb = nhwc(dy)
    ld = Cin + 16
    dxb = torch.zeros((B, H, W, ld), dtype=torch.bfloat16, device="cuda")
    for acc, mult in ((0, 1.0), (1, 2.0)):
        _lib.check(fn(dyb.data_ptr(), pk["d"].data_ptr(), dxb.data_ptr(), B, H, W, ld, 8, Cin,
                      Cout, Cout, 0, acc, None, stream()), "dgrad_s2 " + form)
        got = nchw(dxb)
        _close(got[:, 8:8 + Cin], mult * x.grad, 1e-2 * mult, 3e-2 * mult, "dgrad s2 " + form)
        assert (got[:, :8] == 0).all() and (got[:, 8 + Cin:] == 0).all()


@pytest.mark.parametrize("shape", [(1, 8, 1, 1), (2, 8, 3, 9), (1, 16, 5, 7), (2, 8, 20, 20), (1, 8, 6, 13), (64, 256, 20, 20), (8, 384, 20, 20)])
def test_maxpool_ties_special_values_and_ragged_edges(shape):
    """The key-based 5x5 pool (csrc/misc_ops.hip maxpool5_fwd_kernel) against torch's scan on inputs made of ties and
    special values: small integers (most windows hold their maximum several times: the FIRST one in row-major window
    order must get the gradient), +-inf, and one NaN per image (a NaN beats every number); maps whose sides are not
    multiples of the 4 x 4 output block - and the SPPF's own shapes at full size (yv5s B = 64: 256 channels, yv5m: 384);
    integer-valued output gradients so that the routed sums are exact."""
    B, C, H, W = shape
    g = torch.Generator().manual_seed(H * 31 + W)
    x = torch.randint(-3, 4, (B, C, H, W), generator=g).float()
    x[torch.rand(x.shape, generator=g) < 0.02] = float("inf")
    x[torch.rand(x.shape, generator=g) < 0.05] = float("-inf")
    if H * W >= 9:
        x[:, 0, H // 2, W // 2] = float("nan")
    xr = x.clone().requires_grad_(True)
    y = F.max_pool2d(xr, 5, 1, 2)
    dy = torch.randint(-4, 5, y.shape, generator=g).float()
    y.backward(dy)
    lib = _lib.lib()
    xb = nhwc(bf(x))
    yb = torch.zeros_like(xb)
    idx = torch.zeros((B, H, W, C), dtype=torch.uint8, device="cuda")
    _lib.check(lib.kodhip_maxpool5_fwd(xb.data_ptr(), C, 0, yb.data_ptr(), C, 0, idx.data_ptr(), B, H, W, C, stream()), "pool")
    got = nchw(yb)
    ref = y.detach()
    assert torch.equal(torch.isnan(got), torch.isnan(ref))
    assert torch.equal(torch.nan_to_num(got, nan=7.0), torch.nan_to_num(ref, nan=7.0))
    gb = nhwc(bf(dy))
    dxb = torch.zeros_like(xb)
    _lib.check(lib.kodhip_maxpool5_bwd(gb.data_ptr(), C, 0, idx.data_ptr(), dxb.data_ptr(), C, 0, B, H, W, C, None, stream()), "pool bwd")
    assert torch.equal(nchw(dxb), xr.grad)


def test_conv_channel_slices():
    """Input read from / output written into channel slices of wider buffers (concat elimination)."""
    g = torch.Generator().manual_seed(5)
    B, H, W = 2, 12, 12
    x = bf(torch.randn(B, 64, H, W, generator=g))
    w = bf(torch.randn(32, 32, 3, 3, generator=g) / 17)
    ref = F.conv2d(x[:, 16:48], w, None, 1, 1)
    pk = pack([w])
    xb = nhwc(x)
    out = torch.full((B, H, W, 96), 7.0, dtype=torch.bfloat16, device="cuda")
    conv_fwd_raw(xb, (16, 32), pk, 1, 1, out=out, ycoff=40)
    got = nchw(out)
    _close(got[:, 40:72], ref, 1e-2, 2e-2, "slice conv")
    assert (got[:, :40] == 7).all() and (got[:, 72:] == 7).all()


def test_stem_conv():
    """6x6/s2/p2 stem on a 3-channel image through the pixel-pair layout (forward: wide-pixel FAST form, one 32-value
    K step = four 8-channel pixel pairs per kernel row)."""
    g = torch.Generator().manual_seed(9)
    B, H, W, Cout = 2, 32, 72, 32            # 36 pixel pairs per row: ragged tiles, both horizontal borders
    x = bf(torch.rand(B, 3, H, W, generator=g))
    w = bf(torch.randn(Cout, 3, 6, 6, generator=g) / 10)
    xr, wr = x.clone(), w.clone().requires_grad_(True)
    y = F.conv2d(xr, wr, None, 2, 2)
    dy = bf(torch.randn(y.shape, generator=g))
    y.backward(dy)
    lib = _lib.lib()
    img = torch.empty((B, H, W // 2, 8), dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.kodhip_nchw_to_nhwc4(x.cuda().data_ptr(), img.data_ptr(), B, 3, H, W, stream()), "nhwc4")
    pk = pack([w], stem=True)
    out = torch.zeros((B, H // 2, W // 2, Cout), dtype=torch.bfloat16, device="cuda")
    T = lib.kodhip_conv_stats_slots(B * (H // 2) * (W // 2), Cout)
    stats = torch.zeros(2 * Cout * T, dtype=torch.float32, device="cuda")
    _lib.check(lib.kodhip_conv_fwd_raw(img.data_ptr(), pk["f"].data_ptr(), out.data_ptr(), stats.data_ptr(),
                                       B, H, W // 2, 8, 0, 32, Cout, 6, 1, 2, 1, 2, 1, pk["Kp"], Cout, 0, stream()), "stem")
    _close(nchw(out), y.detach(), 1e-2, 2e-2, "stem fwd")
    dyb = nhwc(dy)
    M = B * (H // 2) * (W // 2)
    Kw = 160                                   # weight-gradient slabs keep the 6x3-tap x 8-channel K (144 -> 160)
    splits = lib.kodhip_conv_wgrad_splits_geo(B, H, W // 2, 8, 8, Cout, 6, 3, 2, 1, 2, 1, Kw, Cout)
    part = torch.zeros(splits * Cout * Kw, dtype=torch.float32, device="cuda")
    gw = torch.zeros_like(w, device="cuda")
    _lib.check(lib.kodhip_conv_wgrad(img.data_ptr(), dyb.data_ptr(), part.data_ptr(), gw.data_ptr(), B, H, W // 2, 8, 0,
                                     8, Cout, 6, 3, 2, 1, 2, 1, Kw, Cout, 0, Cout, 1, 1.0, stream()), "stem wgrad")
    _close(gw.cpu(), wr.grad, 2e-3, 2e-3 * wr.grad.abs().max().item(), "stem wgrad")


@pytest.mark.parametrize("case", [(2, 32, 72, 32, 32, 0), (1, 16, 640, 32, 32, 0), (3, 12, 400, 16, 16, 0), (2, 8, 416, 32, 64, 16),
                                  (2, 16, 400, 48, 48, 0), (1, 12, 640, 64, 96, 24), (2, 8, 72, 40, 40, 0)])
def test_stem_backward_fused(case):
    """kodhip_stem_bwd_fused (BatchNorm/SiLU backward formed inside the stem's weight gradient, dY never written) against
    the two launches it replaces (kodhip_bn_silu_bwd_apply + kodhip_conv_wgrad stem form - same dY rounding, other
    summation order) and against torch on the dY those produce.  Rows of 36 / 320 / 200 / 208 pixel pairs: one ragged
    tile, two full tiles, a full + a ragged tile; 16 output channels; dA as a channel slice of a wider buffer; 48 / 64 / 40
    output channels (yv5m / yv5l stems: two 32-channel tiles per block, the second ragged)."""
    B, H, W, Cout, lda, dacoff = case
    g = torch.Generator().manual_seed(11 + H)
    lib = _lib.lib()
    x = bf(torch.rand(B, 3, H, W, generator=g))
    img = torch.empty((B, H, W // 2, 8), dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.kodhip_nchw_to_nhwc4(x.cuda().data_ptr(), img.data_ptr(), B, 3, H, W, stream()), "nhwc4")
    Ho, Wo = H // 2, W // 2
    M = B * Ho * Wo
    y = torch.randn(M, Cout, generator=g).to(torch.bfloat16).cuda()
    dA = torch.randn(M, lda, generator=g).to(torch.bfloat16).cuda()
    scale = (torch.rand(Cout, generator=g) + 0.5).cuda()
    shift = (torch.randn(Cout, generator=g) * 0.3).cuda()
    coef = torch.cat([torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.05,
                      torch.randn(Cout, generator=g) * 0.05]).cuda()
    # the fused kernel
    blocks = lib.kodhip_stem_bwd_fused_blocks(B, H, Wo, Cout)
    part = torch.full((blocks * (32 if Cout <= 32 else 64) * 160,), float("nan"), dtype=torch.float32, device="cuda")
    gw = torch.zeros((Cout, 3, 6, 6), dtype=torch.float32, device="cuda")
    y_before = y.clone()
    _lib.check(lib.kodhip_stem_bwd_fused(img.data_ptr(), dA.data_ptr(), lda, dacoff, y.data_ptr(), Cout,
                                         scale.data_ptr(), shift.data_ptr(), coef.data_ptr(), part.data_ptr(), gw.data_ptr(),
                                         B, H, Wo, Cout, 1.0, stream()), "stem_bwd_fused")
    torch.cuda.synchronize()
    assert torch.equal(y, y_before), "y must stay untouched"
    # the two launches it replaces
    dy = y.clone()
    _lib.check(lib.kodhip_bn_silu_bwd_apply(dA.data_ptr(), lda, dacoff, dy.data_ptr(), Cout, scale.data_ptr(), shift.data_ptr(),
                                            coef.data_ptr(), None, 0, 0, 0, M, Cout, stream()), "bwd_apply")
    Kw = 160
    splits = lib.kodhip_conv_wgrad_splits_geo(B, H, Wo, 8, 8, Cout, 6, 3, 2, 1, 2, 1, Kw, Cout)
    part2 = torch.zeros(splits * Cout * Kw, dtype=torch.float32, device="cuda")
    gw2 = torch.zeros_like(gw)
    _lib.check(lib.kodhip_conv_wgrad(img.data_ptr(), dy.data_ptr(), part2.data_ptr(), gw2.data_ptr(), B, H, Wo, 8, 0,
                                     8, Cout, 6, 3, 2, 1, 2, 1, Kw, Cout, 0, Cout, 1, 1.0, stream()), "stem wgrad")
    torch.cuda.synchronize()
    top = gw2.abs().max().item()
    _close(gw.cpu(), gw2.cpu(), 1e-4, 2e-5 * top, "fused vs two launches")
    # and torch's convolution weight gradient on that dY
    wr = torch.zeros(Cout, 3, 6, 6, requires_grad=True)
    F.conv2d(x, wr, None, 2, 2).backward(dy.float().cpu().view(B, Ho, Wo, Cout).permute(0, 3, 1, 2))
    _close(gw.cpu(), wr.grad, 2e-3, 2e-3 * wr.grad.abs().max().item(), "fused vs torch")


@pytest.mark.parametrize("case", [(2, 64, 40, 40, 32), (3, 128, 20, 12, 64), (2, 256, 16, 16, 128), (1, 512, 8, 8, 256),
                                  (2, 96, 12, 10, 48), (4, 64, 160, 160, 32)])
def test_conv_wgrad_dual(case):
    """kodhip_conv_wgrad_dual (a CSP layer's main_conv + short_conv weight gradients in one launch over the shared input)
    against fp32 torch and against two kodhip_conv_wgrad launches; x as a channel slice of a wider buffer; every n-tile
    shape (32 / 64 / 128 columns per layer, two tiles per layer for 256), a 48-channel pair (n tiles wider than a layer:
    the first layer's tile must not write into the second layer's slab rows)."""
    B, Cin, H, W, N = case
    g = torch.Generator().manual_seed(sum(case))
    lib = _lib.lib()
    ldx, xcoff = Cin + 32, 16
    xb = torch.randn(B, H, W, ldx, generator=g).to(torch.bfloat16).cuda()
    dy = [torch.randn(B * H * W, N, generator=g).to(torch.bfloat16).cuda() for _ in range(2)]
    Kp = pad(Cin, 32)
    sp = lib.kodhip_conv_wgrad_dual_splits(B, H, W, ldx, Cin, N, Kp, N)
    assert sp > 0
    part = torch.full((sp * 2 * N * Kp,), float("nan"), dtype=torch.float32, device="cuda")
    gw = [torch.zeros(N, Cin, dtype=torch.float32, device="cuda") for _ in range(2)]
    _lib.check(lib.kodhip_conv_wgrad_dual(xb.data_ptr(), dy[0].data_ptr(), dy[1].data_ptr(), part.data_ptr(), gw[0].data_ptr(),
                                          gw[1].data_ptr(), B, H, W, ldx, xcoff, Cin, N, Kp, N, 0, 1.0, stream()), "wgrad_dual")
    torch.cuda.synchronize()
    X = xb[..., xcoff:xcoff + Cin].float().reshape(-1, Cin).cpu().double()
    for i in range(2):
        want = (dy[i].float().cpu().double().t() @ X).float()
        _close(gw[i].cpu(), want, 2e-3, 2e-3 * want.abs().max().item(), f"dual wgrad layer {i} vs torch")
        sp1 = lib.kodhip_conv_wgrad_splits_geo(B, H, W, ldx, Cin, N, 1, 1, 1, 1, 0, 0, Kp, N)
        p1 = torch.zeros(sp1 * N * Kp, dtype=torch.float32, device="cuda")
        g1 = torch.zeros(N, Cin, dtype=torch.float32, device="cuda")
        _lib.check(lib.kodhip_conv_wgrad(xb.data_ptr(), dy[i].data_ptr(), p1.data_ptr(), g1.data_ptr(), B, H, W, ldx, xcoff, Cin,
                                         N, 1, 1, 1, 1, 0, 0, Kp, N, 0, N, 0, 1.0, stream()), "wgrad")
        _close(gw[i].cpu(), g1.cpu(), 1e-4, 2e-5 * want.abs().max().item(), f"dual wgrad layer {i} vs single launch")


def test_head_conv_fwd_bwd():
    g = torch.Generator().manual_seed(3)
    B, C, H, W, A, nc = 2, 128, 8, 8, 3, 10
    x = bf(torch.randn(B, C, H, W, generator=g))
    ws = [bf(torch.randn(n, C, 1, 1, generator=g) / C ** 0.5) for n in (4 * A, A, nc * A)]
    bs = [torch.randn(n, generator=g) for n in (4 * A, A, nc * A)]
    xr = x.clone().requires_grad_(True)
    wr = [w.clone().requires_grad_(True) for w in ws]
    br = [b.clone().requires_grad_(True) for b in bs]
    outs = []
    for w, b, p in zip(wr, br, (4, 1, nc)):
        y = F.conv2d(xr, w, b)
        outs.append(y.view(B, A, p, H, W).permute(0, 1, 3, 4, 2))
    ref = torch.cat(outs, -1)                                   # [B,A,H,W,15]
    gout = torch.randn(ref.shape, generator=g)
    ref.backward(gout)
    lib = _lib.lib()
    npad = pad(A * (5 + nc), 8)
    pk = pack(ws, ntot=npad)
    xb = nhwc(x)
    bias = torch.cat(bs).cuda()
    out = torch.zeros((B, A, H, W, 5 + nc), dtype=torch.float32, device="cuda")
    _lib.check(lib.kodhip_conv_fwd_head(xb.data_ptr(), pk["f"].data_ptr(), bias.data_ptr(), out.data_ptr(),
                                        B, H, W, C, 0, C, A, nc, pk["Kp"], stream()), "head")
    _close(out.cpu(), ref.detach(), 2e-3, 2e-3, "head fwd")
    # backward
    dy = torch.zeros((B * H * W, npad), dtype=torch.bfloat16, device="cuda")
    ws_ = torch.zeros(2048 * npad, dtype=torch.float32, device="cuda")
    db = [torch.zeros(n, device="cuda") for n in (4 * A, A, nc * A)]
    _lib.check(lib.kodhip_head_bwd_prep(gout.cuda().contiguous().data_ptr(), dy.data_ptr(), ws_.data_ptr(),
                                        db[0].data_ptr(), db[1].data_ptr(), db[2].data_ptr(), B, H * W, A, nc, npad,
                                        stream()), "head prep")
    for d, r in zip(db, br):
        _close(d.cpu(), r.grad, 1e-4, 1e-4, "head bias grad")
    dx = torch.zeros((B, H, W, C), dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.kodhip_conv_dgrad(dy.data_ptr(), pk["d"].data_ptr(), dx.data_ptr(), B, H, W, C, 0, C, npad, 1, 1, 1, 1,
                                     0, 0, pk["Kdp"], npad, 0, 0, None, stream()), "head dgrad")
    _close(nchw(dx), xr.grad, 2e-2, 3e-2, "head dgrad")
    splits = lib.kodhip_conv_wgrad_splits_geo(B, H, W, C, C, npad, 1, 1, 1, 1, 0, 0, pk["Kp"], npad)
    part = torch.zeros(splits * npad * pk["Kp"], dtype=torch.float32, device="cuda")
    gw = torch.zeros(A * (5 + nc), C, device="cuda")
    _lib.check(lib.kodhip_conv_wgrad(xb.data_ptr(), dy.data_ptr(), part.data_ptr(), gw.data_ptr(), B, H, W, C, 0, C, npad,
                                     1, 1, 1, 1, 0, 0, pk["Kp"], npad, 0, A * (5 + nc), 0, 1.0, stream()), "head wgrad")
    refw = torch.cat([w.grad.reshape(w.shape[0], C) for w in wr], 0)
    _close(gw.cpu(), refw, 1e-2, 1e-2 * refw.abs().max().item(), "head wgrad")


@pytest.mark.parametrize("C,res,hw", [(32, False, (10, 6)), (64, True, (10, 6)), (48, True, (10, 6)), (256, False, (10, 6)),
                                      # many block chunks with a ragged last one in every launch shape of the apply passes (constants in
                                      # LDS / in registers, one / two rows per thread, line-aligned chunks at 48 and 96 channels)
                                      (32, True, (37, 29)), (48, False, (37, 29)), (96, True, (23, 31)), (128, True, (23, 31))])
def test_bn_silu_fwd_bwd(C, res, hw):
    g = torch.Generator().manual_seed(C)
    B, (H, W) = 3, hw
    M = B * H * W
    y = bf(torch.randn(B, C, H, W, generator=g) * 2 + 0.5)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    ident = bf(torch.randn(B, C, H, W, generator=g)) if res else None
    bn = torch.nn.BatchNorm2d(C, eps=1e-3, momentum=0.03)
    with torch.no_grad():
        bn.weight.copy_(gamma); bn.bias.copy_(beta)
    yr = y.clone().requires_grad_(True)
    ir = ident.clone().requires_grad_(True) if res else None
    out = F.silu(bn(yr))
    if res:
        out = out + ir
    dout = bf(torch.randn(out.shape, generator=g))
    out.backward(dout)
    lib = _lib.lib()
    yb = nhwc(y)
    # statistics from the tensor itself (the conv epilogue is tested separately)
    y2 = yb.float().view(M, C)
    stats = torch.stack((y2.sum(0), (y2 * y2).sum(0))).contiguous().view(2, C, 1)
    sums = torch.zeros(2 * C, dtype=torch.float64, device="cuda")
    _lib.check(lib.kodhip_bn_reduce_partials(stats.data_ptr(), sums.data_ptr(), C, 1, stream()), "reduce")
    aff = torch.zeros(4 * C, device="cuda")
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    gm, bt = gamma.cuda(), beta.cuda()
    a = aff.data_ptr()
    _lib.check(lib.kodhip_bn_finalize(sums.data_ptr(), float(M), gm.data_ptr(), bt.data_ptr(), rm.data_ptr(),
                                      rv.data_ptr(), 0.03, 1e-3, a, a + 4 * C, a + 8 * C, a + 12 * C, C, 1, stream()), "fin")
    _close(rm.cpu(), bn.running_mean, 1e-4, 1e-5, "running_mean")
    _close(rv.cpu(), bn.running_var, 1e-4, 1e-5, "running_var")
    ldo = C + 16
    ob = torch.zeros((B, H, W, ldo), dtype=torch.bfloat16, device="cuda")
    ib = nhwc(ident) if res else None
    _lib.check(lib.kodhip_bn_silu_apply(yb.data_ptr(), C, a, a + 4 * C, ib.data_ptr() if res else None, C, 0,
                                        ob.data_ptr(), ldo, 8, M, C, stream()), "apply")
    _close(nchw(ob)[:, 8:8 + C], out.detach(), 1e-2, 1e-2, "bn+silu fwd")
    # backward
    dob = torch.zeros((B, H, W, ldo), dtype=torch.bfloat16, device="cuda")
    dob[..., 8:8 + C] = nhwc(dout)
    T2 = lib.kodhip_bn_bwd_slots(M, C)
    bpart = torch.zeros(2 * C * T2, device="cuda")
    _lib.check(lib.kodhip_bn_silu_bwd_reduce(dob.data_ptr(), ldo, 8, yb.data_ptr(), C, a, a + 4 * C, a + 8 * C, a + 12 * C,
                                             bpart.data_ptr(), M, C, stream()), "bwd reduce")
    bs = torch.zeros(2 * C, dtype=torch.float64, device="cuda")
    _lib.check(lib.kodhip_bn_reduce_partials(bpart.data_ptr(), bs.data_ptr(), C, T2, stream()), "bwd sums")
    dg, dbt, coef = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(3 * C, device="cuda")
    _lib.check(lib.kodhip_bn_bwd_coeffs(bs.data_ptr(), bs.data_ptr(), float(M), gm.data_ptr(), a + 8 * C, a + 12 * C,
                                        dg.data_ptr(), dbt.data_ptr(), coef.data_ptr(), C, 0, stream()), "coeffs")
    _close(dg.cpu(), bn.weight.grad, 5e-3, 5e-3 * bn.weight.grad.abs().max().item(), "dgamma")
    _close(dbt.cpu(), bn.bias.grad, 5e-3, 5e-3 * bn.bias.grad.abs().max().item(), "dbeta")
    di = torch.ones((B, H, W, C), dtype=torch.bfloat16, device="cuda") if res else None
    _lib.check(lib.kodhip_bn_silu_bwd_apply(dob.data_ptr(), ldo, 8, yb.data_ptr(), C, a, a + 4 * C, coef.data_ptr(),
                                            di.data_ptr() if res else None, C, 0, 1, M, C, stream()), "bwd apply")
    _close(nchw(yb), yr.grad, 2e-2, 2e-2 * yr.grad.abs().max().item(), "dY")
    if res:
        _close(nchw(di), ir.grad + 1.0, 1e-2, 1e-2, "identity grad (accumulated on ones)")


def test_maxpool_chain_and_upsample():
    g = torch.Generator().manual_seed(1)
    B, C, H, W = 2, 16, 9, 11
    x = bf(torch.randn(B, C, H, W, generator=g))
    xr = x.clone().requires_grad_(True)
    pool = torch.nn.MaxPool2d(5, 1, 2)
    y1 = pool(xr); y2 = pool(y1); y3 = pool(y2)
    cat = torch.cat([xr, y1, y2, y3], 1)
    dcat = bf(torch.randn(cat.shape, generator=g))
    cat.backward(dcat)
    lib = _lib.lib()
    buf = torch.zeros((B, H, W, 4 * C), dtype=torch.bfloat16, device="cuda")
    buf[..., :C] = nhwc(x)
    idx = [torch.zeros((B, H, W, C), dtype=torch.uint8, device="cuda") for _ in range(3)]
    for q in range(3):
        _lib.check(lib.kodhip_maxpool5_fwd(buf.data_ptr(), 4 * C, q * C, buf.data_ptr(), 4 * C, (q + 1) * C,
                                           idx[q].data_ptr(), B, H, W, C, stream()), "pool")
    assert torch.equal(nchw(buf), cat.detach())
    gb = nhwc(dcat)
    for q in (2, 1, 0):
        _lib.check(lib.kodhip_maxpool5_bwd(gb.data_ptr(), 4 * C, (q + 1) * C, idx[q].data_ptr(), gb.data_ptr(), 4 * C,
                                           q * C, B, H, W, C, None, stream()), "pool bwd")
    _close(nchw(gb)[:, :C], xr.grad, 2e-2, 5e-3 * xr.grad.abs().max().item(), "pool chain grad")
    # upsample
    up = F.interpolate(xr, scale_factor=2, mode="nearest")
    dup = bf(torch.randn(up.shape, generator=g))
    xr.grad = None
    up.backward(dup)
    xb = nhwc(x)
    ub = torch.zeros((B, 2 * H, 2 * W, C + 8), dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.kodhip_upsample2x_fwd(xb.data_ptr(), C, 0, ub.data_ptr(), C + 8, 8, B, H, W, C, stream()), "up")
    assert torch.equal(nchw(ub)[:, 8:], up.detach())
    dub = torch.zeros_like(ub)
    dub[..., 8:] = nhwc(dup)
    dx = torch.ones((B, H, W, C), dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.kodhip_upsample2x_bwd(dub.data_ptr(), C + 8, 8, dx.data_ptr(), C, 0, 1, B, H, W, C, None, stream()), "up bwd")
    _close(nchw(dx), xr.grad + 1, 1e-2, 2e-2, "upsample grad (accumulate)")


@pytest.mark.parametrize("act,slope", [(1, 0.0), (2, 0.1), (3, 0.0), (4, 0.0), (0, 0.0)])
def test_bn_act_passes_vs_torch(act, slope):
    """kodhip_bn_act_apply / _bwd_reduce / _bwd_apply (activations other than SiLU: 1 ReLU, 2 LeakyReLU, 3 Hardswish,
    4 identity; 0 dispatches to the SiLU kernels) against torch autograd of act(y * scale + shift) (+ residual) on
    bf16-rounded tensors with values on the activations' kinks (0, -3, 3)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(40 + act)
    M, C = 3000, 32
    fn = {0: F.silu, 1: F.relu, 2: lambda t: F.leaky_relu(t, slope), 3: F.hardswish, 4: lambda t: t}[act]
    y = bf(torch.randn(M, C, generator=g) * 2)
    y[::7] = 0.0; y[1::11] = 3.0; y[2::13] = -3.0
    res = bf(torch.randn(M, C, generator=g))
    scale, shift = torch.ones(C), torch.zeros(C)
    scale[::2] = 1.5; shift[::3] = 0.5
    lib = _lib.lib()
    yb, rb = y.to(torch.bfloat16).cuda(), res.to(torch.bfloat16).cuda()
    out = torch.zeros((M, C), dtype=torch.bfloat16, device="cuda")
    sc, sh = scale.cuda(), shift.cuda()
    _lib.check(lib.kodhip_bn_act_apply(yb.data_ptr(), C, sc.data_ptr(), sh.data_ptr(), rb.data_ptr(), C, 0, out.data_ptr(), C, 0, M, C,
                                       act, slope, stream()), "apply")
    yr = y.clone().requires_grad_(True)
    z = yr * scale + shift
    a = fn(z)
    _close(out.float().cpu(), (a + res).detach(), 1e-2, 1e-2, "act(bn(y)) + residual")
    # backward: dz = dA * act'(z); sums and dY = k1 dz + k2 y + k3
    dA = bf(torch.randn(M, C, generator=g))
    a.backward(dA)
    dz = yr.grad / scale                                     # d loss / d z
    mean, rstd = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    T = lib.kodhip_bn_bwd_slots(M, C)
    part = torch.zeros(2 * C * T, device="cuda")
    dAb = dA.to(torch.bfloat16).cuda()
    mean_d, rstd_d = mean.cuda(), rstd.cuda()
    _lib.check(lib.kodhip_bn_act_bwd_reduce(dAb.data_ptr(), C, 0, yb.data_ptr(), C, sc.data_ptr(), sh.data_ptr(), mean_d.data_ptr(),
                                            rstd_d.data_ptr(), part.data_ptr(), M, C, act, slope, stream()), "reduce")
    p2 = part.view(2, C, T).sum(-1).cpu()
    _close(p2[0], dz.sum(0), 2e-3, 2e-3 * dz.abs().sum(0).max().item(), "sum dz")
    _close(p2[1], (dz * (y - mean) * rstd).sum(0), 2e-3, 2e-3 * (dz * (y - mean) * rstd).abs().sum(0).max().item(), "sum dz xhat")
    coef = torch.cat([torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1])
    ybw = yb.clone()
    coef_d = coef.cuda()
    _lib.check(lib.kodhip_bn_act_bwd_apply(dAb.data_ptr(), C, 0, ybw.data_ptr(), C, sc.data_ptr(), sh.data_ptr(), coef_d.data_ptr(),
                                           None, 0, 0, 0, M, C, act, slope, stream()), "bwd apply")
    want = coef[:C] * dz + coef[C:2 * C] * y + coef[2 * C:]
    _close(ybw.float().cpu(), want, 1e-2, 2e-2, "dY")


@pytest.mark.parametrize("K,shape", [(3, (2, 16, 9, 11)), (7, (2, 8, 12, 10)), (9, (1, 24, 20, 20)), (13, (2, 8, 6, 17)), (1, (1, 8, 4, 4)),
                                     (5, (2, 16, 9, 11))])
def test_maxpool_any_window_vs_torch(K, shape):
    """kodhip_maxpool_fwd / _bwd (SPPFBottleneck's kernel_sizes other than 5, kod/nn/layers/sppf.py:27-67): values, the
    argmax routing of the gradient (ties: bf16 values from a small set - torch's first maximum wins) and the accumulation
    into an existing gradient, against torch.nn.MaxPool2d(K, 1, K // 2) on the same bf16 values; K = 5 dispatches to the
    tuned kernels and must agree the same way."""
    B, C, H, W = shape
    g = torch.Generator().manual_seed(K)
    x = bf(torch.randint(-5, 6, (B, C, H, W), generator=g).float() * 0.5)
    xr = x.clone().requires_grad_(True)
    y = torch.nn.MaxPool2d(K, 1, K // 2)(xr)
    dy = bf(torch.randn(y.shape, generator=g))
    y.backward(dy)
    lib = _lib.lib()
    xb, yb = nhwc(x), torch.zeros((B, H, W, C), dtype=torch.bfloat16, device="cuda")
    idx = torch.zeros((B, H, W, C), dtype=torch.uint8, device="cuda")
    _lib.check(lib.kodhip_maxpool_fwd(xb.data_ptr(), C, 0, yb.data_ptr(), C, 0, idx.data_ptr(), B, H, W, C, K, stream()), "pool")
    assert torch.equal(nchw(yb), y.detach())
    base = bf(torch.randn(B, C, H, W, generator=g))
    dxb = nhwc(base)
    _lib.check(lib.kodhip_maxpool_bwd(nhwc(dy).data_ptr(), C, 0, idx.data_ptr(), dxb.data_ptr(), C, 0, B, H, W, C, K, None, stream()), "pool bwd")
    _close(nchw(dxb), base + xr.grad, 2e-2, 1e-2 * (base + xr.grad).abs().max().item(), f"pool {K} gradient")
    assert lib.kodhip_maxpool_fwd(xb.data_ptr(), C, 0, yb.data_ptr(), C, 0, idx.data_ptr(), B, H, W, C, 4, stream()) != 0     # even window: refused


@pytest.mark.parametrize("nesterov,dampening,maximize", [(True, 0.0, False), (False, 0.0, False), (False, 0.3, False),
                                                         (False, 0.0, True), (False, 0.25, True)])
def test_sgd_nesterov(nesterov, dampening, maximize):
    """kodhip_sgd_nesterov against torch.optim.SGD itself: Nesterov momentum (smart_sgd.yaml), plain momentum, dampening
    (the first step copies the gradient into the momentum buffer: flag 4 of hyper[10]) and maximize; three groups with
    their own lr / momentum / weight decay, three steps."""
    g = torch.Generator().manual_seed(2)
    n = 64 * 7
    p = torch.randn(n, generator=g); gr = torch.randn(n, generator=g)
    gid = torch.tensor([0, 1, 2, 1, 255, 0, 2], dtype=torch.uint8)
    lr, mom, wd = (0.1, 0.01, 0.02), (0.8, 0.9, 0.937), (0.0, 5e-4, 0.0)
    pc, buf = p.clone().cuda(), torch.zeros(n, device="cuda")
    lib = _lib.lib()
    grc, gidc = gr.cuda(), gid.cuda()
    refs = [torch.nn.Parameter(p[64 * k:64 * k + 64].clone()) for k in range(7)]
    opt = torch.optim.SGD([dict(params=[refs[k] for k in range(7) if int(gid[k]) == gi], lr=lr[gi], momentum=mom[gi],
                                weight_decay=wd[gi]) for gi in range(3)], lr=0.1, nesterov=nesterov, momentum=0.5,
                          dampening=dampening, maximize=maximize)
    for step in range(3):
        flags = (1.0 if nesterov else 0.0) + (2.0 if maximize else 0.0) + (4.0 if (dampening and step == 0) else 0.0)
        hyper = torch.tensor([*lr, *mom, *wd, 0.5, flags, dampening], dtype=torch.float32, device="cuda")
        _lib.check(lib.kodhip_sgd_nesterov(pc.data_ptr(), grc.data_ptr(), buf.data_ptr(), gidc.data_ptr(), n,
                                           hyper.data_ptr(), stream()), "sgd")
        for k in range(7):
            refs[k].grad = gr[64 * k:64 * k + 64].clone() * 0.5
        opt.step()
    got = pc.cpu()
    for k in range(7):
        want = p[64 * k:64 * k + 64] if int(gid[k]) > 2 else refs[k].detach()          # padding granules stay untouched
        _close(got[64 * k:64 * k + 64], want, 1e-6, 1e-6, f"sgd granule {k}")


@pytest.mark.parametrize("case", [
    # B, Cin (dX channels), H, W, Cout (dY channels), k, s, p, producer channel split of dX
    (2, 64, 24, 20, 64, 3, 1, 1, (64,)),            # one producer owns the whole output
    (2, 128, 16, 12, 32, 1, 1, 0, (64, 64)),        # concat buffer: two producers, short reduction
    (3, 96, 10, 14, 64, 1, 1, 0, (32, 64)),         # ragged split, segment boundary inside a 128-wide tile
    (2, 64, 16, 24, 128, 3, 2, 1, (32, 32)),        # stride-2 form (four parity classes share the slot range)
    (2, 64, 16, 24, 128, 3, 2, 1, (32, 32), "fold"),  # the same through the folded stride-2 form
    (4, 32, 64, 96, 64, 3, 2, 1, (32,), "fold"),
    (4, 128, 72, 64, 128, 3, 1, 1, (128,)),         # M = 18432 rows, K = 1152: 256-pixel tiles
])
def test_conv_dgrad_with_fused_bn_backward_reduction(case):
    """kodhip_conv_dgrad_bnred / _s2_bnred: the data gradient is unchanged, and the per-segment partials (sum dz,
    sum dz*y), turned into coefficients with raw_moment=1, equal the separate reduce pass over the same tensors."""
    from object_detection_cib_amd._lib import KodBnRedSeg
    import ctypes as C
    B, Cin, H, W, Cout, k, s, p, split = case[:9]
    fold = len(case) > 9
    g = torch.Generator().manual_seed(sum(case[:8]))
    x = bf(torch.randn(B, Cin, H, W, generator=g)).requires_grad_(True)
    w = bf(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5)
    y = F.conv2d(x, w, None, s, p)
    dy = bf(torch.randn(y.shape, generator=g))
    y.backward(dy)
    lib = _lib.lib()
    s2 = (k, s, p) == (3, 2, 1)
    pk = pack([w], s2="fold" if fold else s2)
    dyb = nhwc(dy)
    M = B * H * W
    if fold:
        slots = lib.kodhip_conv_dgrad_s2f_bnred_slots(B, H, W, Cin, Cout, Cout)
    else:
        slots = lib.kodhip_conv_dgrad_bnred_slots(B, H, W, Cin, Cout, k, k, s, s, p, p, Cout, int(s2))
    assert slots > 0
    # producers of dX's channel ranges: pre-BN tensors + BN constants
    prods, ch0 = [], 0
    for c in split:
        raw = bf(torch.randn(B, c, H, W, generator=g))
        aff = torch.cat([torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3,      # scale, shift
                         torch.randn(c, generator=g) * 0.2, torch.rand(c, generator=g) + 0.5]).cuda()   # mean, rstd
        prods.append(dict(c=c, ch0=ch0, raw=nhwc(raw), aff=aff,
                          part=torch.full((2 * c * slots,), float("nan"), device="cuda"),
                          gamma=(torch.rand(c, generator=g) + 0.5).cuda()))
        ch0 += c
    segs = (KodBnRedSeg * len(prods))()
    for i, pr in enumerate(prods):
        segs[i].ch_begin, segs[i].ch_count = pr["ch0"], pr["c"]
        segs[i].raw, segs[i].ldr = pr["raw"].data_ptr(), pr["c"]
        segs[i].aff, segs[i].partials = pr["aff"].data_ptr(), pr["part"].data_ptr()
    dxb = torch.zeros((B, H, W, Cin), dtype=torch.bfloat16, device="cuda")
    sp = C.cast(segs, C.c_void_p)
    if s2:
        fn = lib.kodhip_conv_dgrad_s2f_bnred if fold else lib.kodhip_conv_dgrad_s2_bnred
        _lib.check(fn(dyb.data_ptr(), pk["d"].data_ptr(), dxb.data_ptr(), B, H, W, Cin, 0, Cin,
                      Cout, Cout, 0, 0, None, sp, len(prods), slots, stream()), "dgrad_s2_bnred")
    else:
        _lib.check(lib.kodhip_conv_dgrad_bnred(dyb.data_ptr(), pk["d"].data_ptr(), dxb.data_ptr(), B, H, W, Cin, 0, Cin,
                                               Cout, k, k, s, s, p, p, pk["Kdp"], Cout, 0, 0, None, sp, len(prods), slots,
                                               stream()), "dgrad_bnred")
    _close(nchw(dxb), x.grad, 1e-2, 3e-2, "dX of the fused launch")
    for pr in prods:
        c = pr["c"]
        a = pr["aff"].data_ptr()
        out_f = [torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda"), torch.zeros(3 * c, device="cuda")]
        _lib.check(lib.kodhip_bn_bwd_coeffs_partials(pr["part"].data_ptr(), slots, float(M), pr["gamma"].data_ptr(),
                                                     a + 8 * c, a + 12 * c, out_f[0].data_ptr(), out_f[1].data_ptr(),
                                                     out_f[2].data_ptr(), c, 1, stream()), "coeffs fused")
        # the separate pass over the very same (dX slice, raw) tensors
        T2 = lib.kodhip_bn_bwd_slots(M, c)
        bpart = torch.zeros(2 * c * T2, device="cuda")
        _lib.check(lib.kodhip_bn_silu_bwd_reduce(dxb.data_ptr(), Cin, pr["ch0"], pr["raw"].data_ptr(), c, a, a + 4 * c,
                                                 a + 8 * c, a + 12 * c, bpart.data_ptr(), M, c, stream()), "reduce")
        out_s = [torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda"), torch.zeros(3 * c, device="cuda")]
        _lib.check(lib.kodhip_bn_bwd_coeffs_partials(bpart.data_ptr(), T2, float(M), pr["gamma"].data_ptr(),
                                                     a + 8 * c, a + 12 * c, out_s[0].data_ptr(), out_s[1].data_ptr(),
                                                     out_s[2].data_ptr(), c, 0, stream()), "coeffs separate")
        for got, want, what in zip(out_f, out_s, ("dgamma", "dbeta", "coef")):
            assert torch.isfinite(got).all(), what
            _close(got.cpu(), want.cpu(), 2e-4, 2e-4 * want.abs().max().item(), what)


@pytest.mark.parametrize("case", [
    # B, Cin (dX channels), H, W, N (dY channels of EACH of the two convolutions), ld of dX, coff, producer split
    (2, 64, 24, 20, 32, 64, 0, None),
    (3, 128, 20, 12, 64, 192, 64, (64, 64)),       # dX is a channel slice of a concat buffer + fused BN-backward reduction
    (2, 96, 14, 10, 48, 96, 0, (96,)),             # 48 dY channels: the K axis of each source is padded to 64
    (4, 256, 40, 40, 128, 256, 0, None),           # 6400 rows: 256-pixel tiles
])
def test_conv_dgrad_dual_source(case):
    """kodhip_conv_dgrad_dual[_bnred]: dX = dgrad(dY1, W1) + dgrad(dY2, W2) of two pointwise convolutions that read the
    same input (a CSP layer's main_conv / short_conv) in ONE launch; equals the two separate launches' sum and torch."""
    from object_detection_cib_amd._lib import KodBnRedSeg
    import ctypes as C
    B, Cin, H, W, N, ld, coff, split = case
    g = torch.Generator().manual_seed(sum(case[:7]))
    x = bf(torch.randn(B, Cin, H, W, generator=g)).requires_grad_(True)
    ws = [bf(torch.randn(N, Cin, 1, 1, generator=g) / Cin ** 0.5) for _ in range(2)]
    dys = [bf(torch.randn(B, N, H, W, generator=g)) for _ in range(2)]
    for w, dy in zip(ws, dys):
        F.conv2d(x, w).backward(dy)
    lib = _lib.lib()
    pks = [pack([w]) for w in ws]
    dyb = [nhwc(dy) for dy in dys]
    M = B * H * W
    for acc in (0, 1):
        dxb = torch.full((B, H, W, ld), 0.5, dtype=torch.bfloat16, device="cuda")
        if split is None:
            _lib.check(lib.kodhip_conv_dgrad_dual(dyb[0].data_ptr(), pks[0]["d"].data_ptr(), dyb[1].data_ptr(),
                                                  pks[1]["d"].data_ptr(), dxb.data_ptr(), B, H, W, ld, coff, Cin, N,
                                                  pks[0]["Kdp"], N, 0, acc, None, stream()), "dgrad_dual")
        else:
            slots = lib.kodhip_conv_dgrad_dual_bnred_slots(B, H, W, Cin, N, N)
            assert slots > 0
            prods, ch0 = [], 0
            for c in split:
                raw = bf(torch.randn(B, c, H, W, generator=g))
                aff = torch.cat([torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3,
                                 torch.randn(c, generator=g) * 0.2, torch.rand(c, generator=g) + 0.5]).cuda()
                prods.append(dict(c=c, ch0=ch0, raw=nhwc(raw), aff=aff,
                                  part=torch.full((2 * c * slots,), float("nan"), device="cuda")))
                ch0 += c
            segs = (KodBnRedSeg * len(prods))()
            for i, pr in enumerate(prods):
                segs[i].ch_begin, segs[i].ch_count = pr["ch0"], pr["c"]
                segs[i].raw, segs[i].ldr = pr["raw"].data_ptr(), pr["c"]
                segs[i].aff, segs[i].partials = pr["aff"].data_ptr(), pr["part"].data_ptr()
            _lib.check(lib.kodhip_conv_dgrad_dual_bnred(dyb[0].data_ptr(), pks[0]["d"].data_ptr(), dyb[1].data_ptr(),
                                                        pks[1]["d"].data_ptr(), dxb.data_ptr(), B, H, W, ld, coff, Cin, N,
                                                        pks[0]["Kdp"], N, 0, acc, None, C.cast(segs, C.c_void_p), len(prods),
                                                        slots, stream()), "dgrad_dual_bnred")
        got = nchw(dxb)
        _close(got[:, coff:coff + Cin], x.grad + 0.5 * acc, 2e-2, 4e-2, f"dual dgrad acc={acc}")
        rest = torch.cat([got[:, :coff], got[:, coff + Cin:]], 1)
        assert (rest == 0.5).all(), "channels outside the view must stay untouched"
        if split is not None:
            # the fused reduction's coefficients == those of the separate reduce pass over the written dX
            for pr in prods:
                c, a = pr["c"], pr["aff"].data_ptr()
                T2 = lib.kodhip_bn_bwd_slots(M, c)
                bpart = torch.zeros(2 * c * T2, device="cuda")
                _lib.check(lib.kodhip_bn_silu_bwd_reduce(dxb.data_ptr(), ld, coff + pr["ch0"], pr["raw"].data_ptr(), c, a,
                                                         a + 4 * c, a + 8 * c, a + 12 * c, bpart.data_ptr(), M, c,
                                                         stream()), "reduce")
                gamma = torch.ones(c, device="cuda")
                outs = []
                for part, T, rawm in ((pr["part"], slots, 1), (bpart, T2, 0)):
                    o = [torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda"), torch.zeros(3 * c, device="cuda")]
                    _lib.check(lib.kodhip_bn_bwd_coeffs_partials(part.data_ptr(), T, float(M), gamma.data_ptr(), a + 8 * c,
                                                                 a + 12 * c, o[0].data_ptr(), o[1].data_ptr(),
                                                                 o[2].data_ptr(), c, rawm, stream()), "coeffs")
                    outs.append(o)
                for got_, want_, what in zip(outs[0], outs[1], ("dgamma", "dbeta", "coef")):
                    assert torch.isfinite(got_).all(), what
                    _close(got_.cpu(), want_.cpu(), 2e-4, 2e-4 * want_.abs().max().item(), what)


@pytest.mark.parametrize("case", [(4, 16, 32, 32, 16), (4, 64, 8, 8, 128), (6, 32, 16, 16, 32), (4, 128, 4, 4, 128), (2, 48, 9, 7, 96)])
def test_conv3x3_result_independent_of_tile_position(case):
    """The 3x3 / stride-1 kernels (ROW3: a staged row segment shared by the three taps of a kernel row, border taps masked
    per lane) must give every pixel the same reduction whatever tile and tile row it lands in: a batch and its two halves
    agree bit for bit, forward (with BatchNorm statistics vs torch) and data gradient."""
    B, C, H, W, N = case
    lib = _lib.lib()
    g = torch.Generator().manual_seed(sum(case))
    x = bf(torch.randn(B, C, H, W, generator=g))
    w = bf(torch.randn(N, C, 3, 3, generator=g) / (C * 9) ** 0.5)
    dy = bf(torch.randn(B, N, H, W, generator=g))
    pk = pack([w])
    xb, dyb = nhwc(x), nhwc(dy)

    def fwd(xs):
        y, st = conv_fwd_raw(xs, (0, C), pk, 1, 1)
        return y, st.sum(-1)

    def dgrad(ds):
        b = ds.shape[0]
        dx = torch.zeros(b, H, W, C, device="cuda", dtype=torch.bfloat16)
        _lib.check(lib.kodhip_conv_dgrad(ds.data_ptr(), pk["d"].data_ptr(), dx.data_ptr(), b, H, W, C, 0, C, N, 3, 3, 1, 1,
                                         1, 1, pk["Kdp"], N, 0, 0, None, stream()), "dgrad")
        return dx
    y, st = fwd(xb)
    ref = F.conv2d(x, w, None, 1, 1)
    _close(nchw(y), ref, 1e-2, 3e-2, "forward")
    want = torch.stack([ref.sum((0, 2, 3)), (ref * ref).sum((0, 2, 3))])
    _close(st.cpu(), want, 5e-3, 5e-3 * want.abs().max().item(), "batch statistics")
    h = B // 2
    y0, _ = fwd(xb[:h].contiguous())
    y1, _ = fwd(xb[h:].contiguous())
    assert torch.equal(torch.cat([y0, y1]), y), "forward: a pixel's result depends on its tile position"
    dx = dgrad(dyb)
    assert torch.equal(torch.cat([dgrad(dyb[:h].contiguous()), dgrad(dyb[h:].contiguous())]), dx), "dgrad: tile position"


def test_wgrad_row3_form_on_every_eligible_layer():
    """conv_wgrad_row3_kernel is selected only where it measured faster (N <= 32 or Cin >= 256); the other wave layouts stay
    compiled and must stay right: the conv cases again, in a fresh process, with KODHIP_WGRAD_ROW3=2 (every eligible 3x3 /
    stride-1 layer takes the ROW3 form).  (The kernel knobs are read once per process, hence the child process.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KODHIP_WGRAD_ROW3="2")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_hip_ops.py"), "-q", "-x", "-m", "gpu",
                        "-k", "test_conv_fwd_dgrad_wgrad"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert "passed" in r.stdout
