"""The two BatchNorm / SiLU apply passes at the step's tensor sizes (yv5s, B=64): us and GB/s per launch.
A/B knobs: KODHIP_BN_U (rows in flight per thread), KODHIP_BN_GRID (grid cap)."""
import sys, torch
import os; _R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [_R, os.path.join(_R, "tests")]
from object_detection_cib_amd import _lib
from hip_helpers import stream
lib = _lib.lib()
for M, C in ((6553600, 32), (1638400, 64), (1638400, 32), (409600, 128), (409600, 64), (102400, 256), (102400, 128), (25600, 512), (25600, 256)):
    y = torch.randn(M, C, device="cuda").to(torch.bfloat16); out = torch.empty_like(y); dA = torch.randn(M, C, device="cuda").to(torch.bfloat16)
    sc, sh = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"); coef = torch.ones(3 * C, device="cuda")
    def f(): _lib.check(lib.kodhip_bn_silu_apply(y.data_ptr(), C, sc.data_ptr(), sh.data_ptr(), None, 0, 0, out.data_ptr(), C, 0, M, C, stream()))
    def b(): _lib.check(lib.kodhip_bn_silu_bwd_apply(dA.data_ptr(), C, 0, y.data_ptr(), C, sc.data_ptr(), sh.data_ptr(), coef.data_ptr(), None, 0, 0, 0, M, C, stream()))
    line = f"[{M:8d} x {C:3d}]"
    for fn, nb in ((f, 4.0), (b, 6.0)):
        for _ in range(3): fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(10): fn()
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(5): g.replay()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        line += f" | {fn.__name__} {us:7.1f} us {nb * M * C / us / 1e3:6.0f} GB/s"
    print(line, flush=True)
