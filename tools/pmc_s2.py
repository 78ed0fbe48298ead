"""PMC study of the shallow stride-2 data gradients ("say why per launch"): one script that launches, a few
times each, the stage-1 / stage-2 stride-2 units' forward, plain data gradient and data gradient + fused BatchNorm-backward
reduction, and - as the bandwidth reference - the backward BatchNorm / SiLU apply pass over the same dX tensor.  Run under
`rocprofv3 --pmc ... --kernel-trace` (tools/pmc_s2.sh); tools/pmc_s2_reduce.py turns the counter CSVs into a table keyed by
the labels this script prints in dispatch order."""
import ctypes as C
import os
import sys

import torch

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [_R, os.path.join(_R, "tests")]
from object_detection_cib_amd import _lib
from object_detection_cib_amd._lib import KodBnRedSeg
from hip_helpers import pack, stream

lib = _lib.lib()
B, REP = 64, 4
for name, Cin, H, Cout in (("s1 32->64 @320", 32, 320, 64), ("s2 64->128 @160", 64, 160, 128), ("s3 128->256 @80", 128, 80, 256)):
    W = H
    g = torch.Generator().manual_seed(1)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    dy = torch.randn(B, H // 2, W // 2, Cout, device="cuda").to(torch.bfloat16)
    dx = torch.empty(B, H, W, Cin, device="cuda", dtype=torch.bfloat16)
    raw = torch.randn(B, H, W, Cin, device="cuda").to(torch.bfloat16)
    x = torch.randn(B, H, W, Cin, device="cuda").to(torch.bfloat16)
    yo = torch.empty(B, H // 2, W // 2, Cout, device="cuda", dtype=torch.bfloat16)
    aff = torch.cat([torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.3, torch.randn(Cin) * 0.2, torch.rand(Cin) + 0.5]).cuda()
    coef = torch.ones(3 * Cin, device="cuda")
    st = stream()
    fold = bool(lib.kodhip_conv_dgrad_s2_folded(Cin, Cout))
    pk, pkf = pack([w], s2="fold" if fold else True), pack([w])
    if fold:
        slots = lib.kodhip_conv_dgrad_s2f_bnred_slots(B, H, W, Cin, Cout, Cout)
        fn, fnp = lib.kodhip_conv_dgrad_s2f_bnred, lib.kodhip_conv_dgrad_s2f
    else:
        slots = lib.kodhip_conv_dgrad_bnred_slots(B, H, W, Cin, Cout, 3, 3, 2, 2, 1, 1, Cout, 1)
        fn, fnp = lib.kodhip_conv_dgrad_s2_bnred, lib.kodhip_conv_dgrad_s2
    part = torch.zeros(2 * Cin * max(slots, 1), device="cuda")
    segs = (KodBnRedSeg * 1)()
    segs[0].ch_begin, segs[0].ch_count, segs[0].raw, segs[0].ldr = 0, Cin, raw.data_ptr(), Cin
    segs[0].aff, segs[0].partials = aff.data_ptr(), part.data_ptr()
    sp = C.cast(segs, C.c_void_p)
    T = lib.kodhip_conv_stats_slots(B * (H // 2) * (W // 2), Cout)
    stats = torch.zeros(2 * Cout * T, device="cuda")
    torch.cuda.synchronize()
    for _ in range(REP):
        _lib.check(fn(dy.data_ptr(), pk["d"].data_ptr(), dx.data_ptr(), B, H, W, Cin, 0, Cin, Cout, Cout, 0, 0, None, sp, 1, slots, st))
    for _ in range(REP):
        _lib.check(fnp(dy.data_ptr(), pk["d"].data_ptr(), dx.data_ptr(), B, H, W, Cin, 0, Cin, Cout, Cout, 0, 0, None, st))
    for _ in range(REP):
        _lib.check(lib.kodhip_conv_fwd_raw(x.data_ptr(), pkf["f"].data_ptr(), yo.data_ptr(), stats.data_ptr(), B, H, W, Cin, 0, Cin,
                                           Cout, 3, 3, 2, 2, 1, 1, pkf["Kp"], Cout, 0, st))
    for _ in range(REP):
        _lib.check(lib.kodhip_bn_silu_bwd_apply(dx.data_ptr(), Cin, 0, raw.data_ptr(), Cin, aff.data_ptr(), aff.data_ptr() + 4 * Cin,
                                                coef.data_ptr(), None, 0, 0, 0, B * H * W, Cin, st))
    torch.cuda.synchronize()
    mb = lambda *ts: sum(t.numel() * 2 for t in ts) / 1e6
    print(f"LABEL {name} | dgrad+bnred {mb(dy, dx, raw):.0f} MB | dgrad {mb(dy, dx):.0f} MB | fwd {mb(x, yo):.0f} MB | bwd_apply {mb(dx, raw, raw):.0f} MB", flush=True)
