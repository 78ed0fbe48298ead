"""3x3/s2 data gradient: parity-class form vs folded form on the six stride-2 layers of yv5s (B=64, 640 px)."""
import sys, torch
import os; _R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [_R, os.path.join(_R, "tests")]
from object_detection_cib_amd import _lib
from hip_helpers import pack, stream

lib = _lib.lib()
LAYERS = [("s1.conv 32->64 @320", 32, 320, 64), ("s2.conv 64->128 @160", 64, 160, 128), ("s3.conv 128->256 @80", 128, 80, 256),
          ("s4.conv 256->512 @40", 256, 40, 512), ("neck.down0 128->128 @80", 128, 80, 128), ("neck.down1 256->256 @40", 256, 40, 256),
          # yv5m widths (BASELINE configs[4])
          ("m.s1.conv 48->96 @320", 48, 320, 96), ("m.s2.conv 96->192 @160", 96, 160, 192), ("m.s3.conv 192->384 @80", 192, 80, 384),
          ("m.s4.conv 384->768 @40", 384, 40, 768), ("m.down0 192->192 @80", 192, 80, 192), ("m.down1 384->384 @40", 384, 40, 384)]
if len(sys.argv) > 1:
    LAYERS = [l for l in LAYERS if any(a in l[0] for a in sys.argv[1:])]
B = 64
for name, Cin, H, Cout in LAYERS:
    W = H
    w = torch.randn(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5
    dy = torch.randn(B, H // 2, W // 2, Cout, device="cuda").to(torch.bfloat16)
    dx = torch.empty(B, H, W, Cin, device="cuda", dtype=torch.bfloat16)
    st = stream()
    byts = 2.0 * (B * H * W * Cin + dy.numel())
    line = f"{name:28s}"
    outs = []
    for form, fn in (("classes", lib.kodhip_conv_dgrad_s2), ("folded", lib.kodhip_conv_dgrad_s2f)):
        pk = pack([w], s2=True if form == "classes" else "fold")
        call = lambda: _lib.check(fn(dy.data_ptr(), pk["d"].data_ptr(), dx.data_ptr(), B, H, W, Cin, 0, Cin, Cout, Cout, 0, 0, None, st))
        for _ in range(3): call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): call()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        outs.append(dx.clone())
        line += f" | {form:7s} {us:7.1f}us {byts / us / 1e3:6.0f}GB/s"
    line += f" | max diff {(outs[0].float() - outs[1].float()).abs().max().item():.3g}"
    print(line)
