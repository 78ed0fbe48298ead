"""The six 3x3 / stride-2 data gradients of yv5s (B=64, 640 px) WITH their fused BatchNorm-backward reduction, as the step
launches them (folded form for Cin <= 64, parity classes above), next to the same launch without the reduction and next
to the layer's forward.  KODHIP_LIB=tools/ablate/lib_X.so times an ablation build (tools/build_ablate.sh)."""
import ctypes as C
import os
import sys

import torch

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [_R, os.path.join(_R, "tests")]
from object_detection_cib_amd import _lib
from object_detection_cib_amd._lib import KodBnRedSeg
from hip_helpers import pack, stream, conv_fwd_raw

lib = _lib.lib()
LAYERS = [("s1 32->64 @320", 32, 320, 64), ("s2 64->128 @160", 64, 160, 128), ("s3 128->256 @80", 128, 80, 256),
          ("s4 256->512 @40", 256, 40, 512), ("down0 128->128 @80", 128, 80, 128), ("down1 256->256 @40", 256, 40, 256),
          ("m.s1 48->96 @320", 48, 320, 96), ("m.s2 96->192 @160", 96, 160, 192), ("m.s3 192->384 @80", 192, 80, 384),
          ("m.s4 384->768 @40", 384, 40, 768), ("m.down0 192->192 @80", 192, 80, 192), ("m.down1 384->384 @40", 384, 40, 384)]
sel = [a for a in sys.argv[1:] if not a.startswith("-")]
LAYERS = [l for l in LAYERS if (any(a in l[0] for a in sel) if sel else not l[0].startswith("m."))]
B = 64


def timed(call, n=20):
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for name, Cin, H, Cout in LAYERS:
    W = H
    g = torch.Generator().manual_seed(1)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    dy = torch.randn(B, H // 2, W // 2, Cout, device="cuda").to(torch.bfloat16)
    dx = torch.empty(B, H, W, Cin, device="cuda", dtype=torch.bfloat16)
    raw = torch.randn(B, H, W, Cin, device="cuda").to(torch.bfloat16)
    aff = torch.cat([torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.3, torch.randn(Cin) * 0.2, torch.rand(Cin) + 0.5]).cuda()
    st = stream()
    fold = bool(lib.kodhip_conv_dgrad_s2_folded(Cin, Cout))
    pk = pack([w], s2="fold" if fold else True)
    if fold:
        slots = lib.kodhip_conv_dgrad_s2f_bnred_slots(B, H, W, Cin, Cout, Cout)
        fn, fnp = lib.kodhip_conv_dgrad_s2f_bnred, lib.kodhip_conv_dgrad_s2f
    else:
        slots = lib.kodhip_conv_dgrad_bnred_slots(B, H, W, Cin, Cout, 3, 3, 2, 2, 1, 1, Cout, 1)
        fn, fnp = lib.kodhip_conv_dgrad_s2_bnred, lib.kodhip_conv_dgrad_s2
    part = torch.zeros(2 * Cin * max(slots, 1), device="cuda")
    segs = (KodBnRedSeg * 1)()
    segs[0].ch_begin, segs[0].ch_count = 0, Cin
    segs[0].raw, segs[0].ldr = raw.data_ptr(), Cin
    segs[0].aff, segs[0].partials = aff.data_ptr(), part.data_ptr()
    sp = C.cast(segs, C.c_void_p)
    t_bn = timed(lambda: _lib.check(fn(dy.data_ptr(), pk["d"].data_ptr(), dx.data_ptr(), B, H, W, Cin, 0, Cin, Cout, Cout, 0, 0, None,
                                       sp, 1, slots, st)))
    t_pl = timed(lambda: _lib.check(fnp(dy.data_ptr(), pk["d"].data_ptr(), dx.data_ptr(), B, H, W, Cin, 0, Cin, Cout, Cout, 0, 0, None, st)))
    pkf = pack([w])
    x = torch.randn(B, H, W, Cin, device="cuda").to(torch.bfloat16)
    yo = torch.empty(B, H // 2, W // 2, Cout, device="cuda", dtype=torch.bfloat16)
    t_fw = timed(lambda: conv_fwd_raw(x, (0, Cin), pkf, 2, 1, out=yo))
    b_pl = 2.0 * (dx.numel() + dy.numel())
    b_bn = b_pl + 2.0 * raw.numel()
    print(f"{name:20s} {'fold' if fold else 'cls4'} | fwd {t_fw:6.1f} us | dgrad {t_pl:6.1f} us {b_pl / t_pl / 1e3:5.0f} GB/s | "
          f"dgrad+bnred {t_bn:6.1f} us {b_bn / t_bn / 1e3:5.0f} GB/s | x fwd {t_bn / t_fw:4.2f}", flush=True)
