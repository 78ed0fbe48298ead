"""Micro-benchmark of the step's small kernels at their yv5s B=64 / 640 px shapes (SPPF pools, upsample): us per launch."""
import sys, torch
import os; _R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [_R, os.path.join(_R, "tests")]
from object_detection_cib_amd import _lib
from hip_helpers import stream

lib = _lib.lib()
st = stream()


def timed(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for name, (B, H, W, C) in {"sppf pool yv5s [64,20,20,256]": (64, 20, 20, 256), "sppf pool yv5m [64,20,20,384]": (64, 20, 20, 384)}.items():
    buf = torch.randn(B, H, W, 4 * C, device="cuda").to(torch.bfloat16)
    idx = torch.zeros((B, H, W, C), dtype=torch.uint8, device="cuda")
    g = torch.randn(B, H, W, 4 * C, device="cuda").to(torch.bfloat16)
    f = timed(lambda: _lib.check(lib.kodhip_maxpool5_fwd(buf.data_ptr(), 4 * C, 0, buf.data_ptr(), 4 * C, C, idx.data_ptr(), B, H, W, C, st)))
    b = timed(lambda: _lib.check(lib.kodhip_maxpool5_bwd(g.data_ptr(), 4 * C, C, idx.data_ptr(), g.data_ptr(), 4 * C, 0, B, H, W, C, None, st)))
    mb = B * H * W * C * 2 / 1e6
    print(f"{name:34s} | fwd {f:6.1f} us ({(2 * mb + mb / 2) / f * 1e3:6.0f} GB/s) | bwd {b:6.1f} us ({(3 * mb + mb / 2) / b * 1e3:6.0f} GB/s)")

# the heads' gradient re-layout (csrc/misc_ops.hip head_bwd_prep_kernel + bias reduction) at the three levels, nc = 10
for name, HW in {"head grads 80x80": 6400, "head grads 40x40": 1600, "head grads 20x20": 400}.items():
    B, A, nc, Npad = 64, 3, 10, 48
    g = torch.randn(B, A, HW, 5 + nc, device="cuda")
    dy = torch.empty(B * HW, Npad, device="cuda", dtype=torch.bfloat16)
    ws = torch.empty(2048 * Npad, device="cuda")
    db = [torch.empty(n, device="cuda") for n in (4 * A, A, nc * A)]
    t = timed(lambda: _lib.check(lib.kodhip_head_bwd_prep(g.data_ptr(), dy.data_ptr(), ws.data_ptr(), db[0].data_ptr(), db[1].data_ptr(),
                                                        db[2].data_ptr(), B, HW, A, nc, Npad, st)))
    mb = (g.numel() * 4 + dy.numel() * 2) / 1e6
    print(f"{name:34s} | prep + bias reduce {t:6.1f} us ({mb / t * 1e3:6.0f} GB/s)")
