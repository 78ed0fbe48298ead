"""Micro-benchmark of the conv kernels on representative yv5s layers (B=64, 640 px)."""
import sys, torch
import os; _R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [_R, os.path.join(_R, "tests")]
from object_detection_cib_amd import _lib
from hip_helpers import pack, stream

lib = _lib.lib()
LAYERS = {  # name: (Cin, H, W, Cout, k, s, p)
    "s3.b.conv2 128->128 3x3 @40": (128, 40, 40, 128, 3, 1, 1),
    "s2.b.conv2 64->64 3x3 @80": (64, 80, 80, 64, 3, 1, 1),
    "s1.b.conv2 32->32 3x3 @160": (32, 160, 160, 32, 3, 1, 1),
    "s4.b.conv2 256->256 3x3 @20": (256, 20, 20, 256, 3, 1, 1),
    "s2.conv 64->128 3x3s2 @160": (64, 160, 160, 128, 3, 2, 1),
    "s4.conv 256->512 3x3s2 @40": (256, 40, 40, 512, 3, 2, 1),
    "s1.conv 32->64 3x3s2 @320": (32, 320, 320, 64, 3, 2, 1),
    "s1.main 64->32 1x1 @160": (64, 160, 160, 32, 1, 1, 0),
    "s2.last 128->128 1x1 @80": (128, 80, 80, 128, 1, 1, 0),
    "s4.last 512->512 1x1 @20": (512, 20, 20, 512, 1, 1, 0),
    "sppf.conv2 1024->512 1x1 @20": (1024, 20, 20, 512, 1, 1, 0),
    # the deep pointwise layers (strides 16 / 32: M = 102 400 / 25 600 rows - one round of blocks, latency-bound K loops)
    "s3.main 256->128 1x1 @40": (256, 40, 40, 128, 1, 1, 0),
    "s3.conv1 128->128 1x1 @40": (128, 40, 40, 128, 1, 1, 0),
    "s3.last 256->256 1x1 @40": (256, 40, 40, 256, 1, 1, 0),
    "s4.main 512->256 1x1 @20": (512, 20, 20, 256, 1, 1, 0),
    "s4.conv1 256->256 1x1 @20": (256, 20, 20, 256, 1, 1, 0),
    "td0.main 512->128 1x1 @40": (512, 40, 40, 128, 1, 1, 0),
    # yv5m widths (BASELINE configs[4]): the 3x3 layers whose weight gradients are 2.2 - 2.9 x their forward
    "m.s3.b.conv2 192->192 3x3 @40": (192, 40, 40, 192, 3, 1, 1),
    "m.s4.b.conv2 384->384 3x3 @20": (384, 20, 20, 384, 3, 1, 1),
    "m.s2.b.conv2 96->96 3x3 @80": (96, 80, 80, 96, 3, 1, 1),
    "m.s1.b.conv2 48->48 3x3 @160": (48, 160, 160, 48, 3, 1, 1),
    "m.s4.conv 384->768 3x3s2 @40": (384, 40, 40, 768, 3, 2, 1),
    "m.s1.conv 48->96 3x3s2 @320": (48, 320, 320, 96, 3, 2, 1),
    "m.s1.last 96->96 1x1 @160": (96, 160, 160, 96, 1, 1, 0),
    "m.s2.main 192->96 1x1 @80": (192, 80, 80, 96, 1, 1, 0),
    "m.s2.conv 96->192 3x3s2 @160": (96, 160, 160, 192, 3, 2, 1),
    "m.s3.main 384->192 1x1 @40": (384, 40, 40, 192, 1, 1, 0),
    "m.s3.conv1 192->192 1x1 @40": (192, 40, 40, 192, 1, 1, 0),
}
B = 64
which = sys.argv[1:] or list(LAYERS)
for name in which:
    Cin, H, W, Cout, k, s, p = LAYERS[name]
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    M = B * Ho * Wo
    x = torch.randn(B, H, W, Cin, device="cuda").to(torch.bfloat16)
    w = torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5
    s2 = (k == 3 and s == 2)
    fold = s2 and lib.kodhip_conv_dgrad_s2_folded(Cin, Cout) == 1      # the form the engine packs and calls for this layer
    pk = pack([w], s2=("fold" if fold else s2))
    y = torch.empty(B, Ho, Wo, Cout, device="cuda", dtype=torch.bfloat16)
    dy = torch.randn(B, Ho, Wo, Cout, device="cuda").to(torch.bfloat16)
    dx = torch.empty_like(x)
    T = lib.kodhip_conv_stats_slots(M, Cout)
    stats = torch.empty(2 * Cout * T, device="cuda")
    splits = lib.kodhip_conv_wgrad_splits_geo(B, H, W, Cin, Cin, Cout, k, k, s, s, p, p, pk["Kp"], Cout)
    part = torch.empty(splits * Cout * pk["Kp"], device="cuda")
    gw = torch.empty(Cout, Cin, k, k, device="cuda")
    st = stream()
    def fwd(): _lib.check(lib.kodhip_conv_fwd_raw(x.data_ptr(), pk["f"].data_ptr(), y.data_ptr(), stats.data_ptr(), B, H, W, Cin, 0, Cin, Cout, k, k, s, s, p, p, pk["Kp"], Cout, 0, st))
    def dgrad():
        if fold: _lib.check(lib.kodhip_conv_dgrad_s2f(dy.data_ptr(), pk["d"].data_ptr(), dx.data_ptr(), B, H, W, Cin, 0, Cin, Cout, Cout, 0, 0, None, st))
        elif s2: _lib.check(lib.kodhip_conv_dgrad_s2(dy.data_ptr(), pk["d"].data_ptr(), dx.data_ptr(), B, H, W, Cin, 0, Cin, Cout, Cout, 0, 0, None, st))
        else: _lib.check(lib.kodhip_conv_dgrad(dy.data_ptr(), pk["d"].data_ptr(), dx.data_ptr(), B, H, W, Cin, 0, Cin, Cout, k, k, s, s, p, p, pk["Kdp"], Cout, 0, 0, None, st))
    def wgrad(): _lib.check(lib.kodhip_conv_wgrad(x.data_ptr(), dy.data_ptr(), part.data_ptr(), gw.data_ptr(), B, H, W, Cin, 0, Cin, Cout, k, k, s, s, p, p, pk["Kp"], Cout, 0, Cout, 0, 1.0, st))
    flops = 2.0 * M * Cout * Cin * k * k
    byts = 2.0 * (B * H * W * Cin + M * Cout)
    line = f"{name:32s}"
    import os
    only = os.environ.get("BENCH_CONV_ONLY")
    for fn in ((wgrad,) if only == "wgrad" else ((fwd, dgrad) if only == "fd" else (fwd, dgrad, wgrad))):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        line += f" | {fn.__name__:5s} {us:7.1f}us {flops / us / 1e6:6.0f}TF {byts / us / 1e3:6.0f}GB/s"
    print(line)
