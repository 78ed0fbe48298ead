"""Tile-shape sweep of the conv kernel on the deep (small-M) layers of yv5s (B=64, 640 px): run once per
KODHIP_FORCE_BM / KODHIP_FORCE_BN setting (the library reads them once per process); prints one line per layer."""
import os, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from object_detection_cib_amd import _lib
from hip_helpers import pack, stream

lib = _lib.lib()
LAYERS = [  # name, Cin, H, Cout, k
    ("s3.main 256->128 1x1 @40", 256, 40, 128, 1), ("s3.b.conv1 128->128 1x1 @40", 128, 40, 128, 1),
    ("s3.b.conv2 128->128 3x3 @40", 128, 40, 128, 3), ("s3.last 256->256 1x1 @40", 256, 40, 256, 1),
    ("s4.main 512->256 1x1 @20", 512, 20, 256, 1), ("s4.b.conv1 256->256 1x1 @20", 256, 20, 256, 1),
    ("s4.b.conv2 256->256 3x3 @20", 256, 20, 256, 3), ("s4.last 512->512 1x1 @20", 512, 20, 512, 1),
    ("sppf.conv2 1024->512 1x1 @20", 1024, 20, 512, 1), ("td0.main 512->128 1x1 @40", 512, 40, 128, 1),
    ("td1.main 256->64 1x1 @80", 256, 80, 64, 1), ("s2.b.conv1 64->64 1x1 @80", 64, 80, 64, 1),
    ("s2.last 128->128 1x1 @80", 128, 80, 128, 1),
]
B = 64
tag = "bm%s bn%s" % (os.environ.get("KODHIP_FORCE_BM", "-"), os.environ.get("KODHIP_FORCE_BN", "-"))
for name, Cin, H, Cout, k in LAYERS:
    p = k // 2
    M = B * H * H
    x = torch.randn(B, H, H, Cin, device="cuda").to(torch.bfloat16)
    w = torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5
    pk = pack([w])
    y = torch.empty(B, H, H, Cout, device="cuda", dtype=torch.bfloat16)
    dy = torch.randn(B, H, H, Cout, device="cuda").to(torch.bfloat16)
    dx = torch.empty_like(x)
    T = lib.kodhip_conv_stats_slots(M, Cout)
    stats = torch.empty(2 * Cout * T, device="cuda")
    st = stream()
    def fwd(): _lib.check(lib.kodhip_conv_fwd_raw(x.data_ptr(), pk["f"].data_ptr(), y.data_ptr(), stats.data_ptr(), B, H, H, Cin, 0, Cin, Cout, k, k, 1, 1, p, p, pk["Kp"], Cout, 0, st))
    def dgrad(): _lib.check(lib.kodhip_conv_dgrad(dy.data_ptr(), pk["d"].data_ptr(), dx.data_ptr(), B, H, H, Cin, 0, Cin, Cout, k, k, 1, 1, p, p, pk["Kdp"], Cout, 0, 0, None, st))
    line = f"{tag:12s} {name:30s}"
    for fn in (fwd, dgrad):
        try:
            for _ in range(3): fn()
        except RuntimeError:              # this tile shape does not exist for the layer
            line += f" | {fn.__name__:5s}     n/a  "
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(30): fn()
        e1.record(); torch.cuda.synchronize()
        line += f" | {fn.__name__:5s} {e0.elapsed_time(e1) * 1e3 / 30:7.1f}us"
    print(line)
