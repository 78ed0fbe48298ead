"""The two BatchNorm / SiLU apply passes at the step's tensor sizes (yv5s, B=64): us and GB/s per launch.
Successive launches walk through distinct slices of 419 MB allocations, so that a tensor smaller than the Infinity Cache is as
cold as it is inside a training step (BENCH_BN_HOT=1: the same slice every launch, the round-5 form).
A/B knobs (csrc/bn_act.hip apply_shape): KODHIP_BN_U (rows in flight per thread), KODHIP_BN_GRID (grid cap), KODHIP_BN_BLOCK
(threads per block), KODHIP_BN_LDS (1 / -1: constants through LDS or not).  BENCH_BN_SHAPES=yv5m: the yv5m widths."""
import sys, torch
import os; _R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [_R, os.path.join(_R, "tests")]
from object_detection_cib_amd import _lib
from hip_helpers import stream
lib = _lib.lib()
W = 1.5 if os.environ.get("BENCH_BN_SHAPES") == "yv5m" else 1.0      # yv5m: 48 / 96 / 192 / 384 / 768 channels
MAXEL = int(6553600 * 32 * W)
Y = torch.randn(MAXEL, device="cuda").to(torch.bfloat16); OUT = torch.empty_like(Y); DA = torch.randn(MAXEL, device="cuda").to(torch.bfloat16)
hot = bool(os.environ.get("BENCH_BN_HOT"))
tot_f = tot_b = 0.0
SHAPES = ((6553600, 32), (1638400, 64), (1638400, 32), (409600, 128), (409600, 64), (102400, 256), (102400, 128), (25600, 512), (25600, 256))
if os.environ.get("BENCH_BN_EXTRA"):       # "M,C;M,C": other shapes (already at their own widths)
    SHAPES = tuple(tuple(int(v) for v in p.split(",")) for p in os.environ["BENCH_BN_EXTRA"].split(";")); W = 1.0
    MAXEL = max(m * c for m, c in SHAPES)
    Y = torch.randn(MAXEL, device="cuda").to(torch.bfloat16); OUT = torch.empty_like(Y); DA = torch.randn(MAXEL, device="cuda").to(torch.bfloat16)
for M, C in SHAPES:
    C = int(C * W)
    el = M * C
    nrot = 1 if hot else min(8, MAXEL // el)
    sc, sh = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"); coef = torch.ones(3 * C, device="cuda")
    def f(r):
        o = (r % nrot) * el * 2
        _lib.check(lib.kodhip_bn_silu_apply(Y.data_ptr() + o, C, sc.data_ptr(), sh.data_ptr(), None, 0, 0, OUT.data_ptr() + o, C, 0, M, C, stream()))
    def b(r):
        o = (r % nrot) * el * 2
        _lib.check(lib.kodhip_bn_silu_bwd_apply(DA.data_ptr() + o, C, 0, Y.data_ptr() + o, C, sc.data_ptr(), sh.data_ptr(), coef.data_ptr(), None, 0, 0, 0, M, C, stream()))
    line = f"[{M:8d} x {C:3d}]"
    for fn, nb in ((f, 4.0), (b, 6.0)):
        for r in range(3): fn(r)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for r in range(16): fn(r)
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(4): g.replay()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 64
        line += f" | {fn.__name__} {us:7.1f} us {nb * M * C / us / 1e3:6.0f} GB/s"
        if fn is f: tot_f += us
        else: tot_b += us
    print(line, flush=True)
print(f"sum over the sizes: f {tot_f:.1f} us, b {tot_b:.1f} us", flush=True)
