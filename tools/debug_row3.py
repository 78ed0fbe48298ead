"""ROW3 debugging: conv forward / dgrad of a batch vs the same images processed in two half batches (must be bitwise
equal: a pixel's reduction order does not depend on the tile it lands in), and vs torch."""
import sys, torch
import torch.nn.functional as F
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from object_detection_cib_amd import _lib
from hip_helpers import pack, stream, nhwc, nchw, bf
lib = _lib.lib()

def fwd(xb, pk, B, H, W, Cin, Cout):
    y = torch.zeros(B, H, W, Cout, device="cuda", dtype=torch.bfloat16)
    T = lib.kodhip_conv_stats_slots(B * H * W, Cout)
    st = torch.zeros(2 * Cout * T, device="cuda")
    _lib.check(lib.kodhip_conv_fwd_raw(xb.data_ptr(), pk["f"].data_ptr(), y.data_ptr(), st.data_ptr(), B, H, W, Cin, 0, Cin, Cout, 3, 3, 1, 1, 1, 1, pk["Kp"], Cout, 0, stream()))
    return y, st.view(2, Cout, T).sum(-1)

def dgrad(dyb, pk, B, H, W, Cin, Cout):
    dx = torch.zeros(B, H, W, Cin, device="cuda", dtype=torch.bfloat16)
    _lib.check(lib.kodhip_conv_dgrad(dyb.data_ptr(), pk["d"].data_ptr(), dx.data_ptr(), B, H, W, Cin, 0, Cin, Cout, 3, 3, 1, 1, 1, 1, pk["Kdp"], Cout, 0, 0, stream()))
    return dx

for (B, C, H, W, N) in [(2, 16, 32, 32, 16), (2, 32, 16, 16, 32), (2, 64, 8, 8, 64), (6, 16, 32, 32, 16), (4, 16, 32, 32, 16), (4, 32, 16, 16, 32), (4, 64, 8, 8, 64), (4, 128, 4, 4, 128), (4, 64, 8, 8, 128), (2, 128, 4, 4, 128), (4, 16, 16, 16, 16)]:
    g = torch.Generator().manual_seed(B + C + H)
    x = bf(torch.randn(B, C, H, W, generator=g))
    w = bf(torch.randn(N, C, 3, 3, generator=g) / (C * 9) ** 0.5)
    dy = bf(torch.randn(B, N, H, W, generator=g))
    pk = pack([w])
    xb, dyb = nhwc(x), nhwc(dy)
    y, st = fwd(xb, pk, B, H, W, C, N)
    ref = F.conv2d(x, w, None, 1, 1)
    e = (nchw(y) - ref).abs().max().item() / ref.abs().max().item()
    st_ref = torch.stack([ref.sum((0, 2, 3)), (ref * ref).sum((0, 2, 3))])
    es = ((st.cpu() - st_ref).abs().max() / st_ref.abs().max()).item()
    h = B // 2
    y0, _ = fwd(xb[:h].contiguous(), pk, h, H, W, C, N)
    y1, _ = fwd(xb[h:].contiguous(), pk, h, H, W, C, N)
    same = torch.equal(torch.cat([y0, y1]), y)
    dx = dgrad(dyb, pk, B, H, W, C, N)
    xr = x.clone().requires_grad_(True)
    F.conv2d(xr, w, None, 1, 1).backward(dy)
    ed = (nchw(dx) - xr.grad).abs().max().item() / xr.grad.abs().max().item()
    d0 = dgrad(dyb[:h].contiguous(), pk, h, H, W, C, N); d1 = dgrad(dyb[h:].contiguous(), pk, h, H, W, C, N)
    y0, s0 = fwd(xb[:h].contiguous(), pk, h, H, W, C, N)
    r0 = F.conv2d(x[:h], w, None, 1, 1)
    s0r = torch.stack([r0.sum((0, 2, 3)), (r0 * r0).sum((0, 2, 3))])
    es0 = ((s0.cpu() - s0r).abs().max() / s0r.abs().max()).item()
    print(f"   half-batch stats err {es0:.2e}", end="")
    print(f"B{B} C{C} {H}x{W} N{N}: fwd err {e:.2e} stats err {es:.2e} halves equal {same} | dgrad err {ed:.2e} halves equal {torch.equal(torch.cat([d0, d1]), dx)}")
