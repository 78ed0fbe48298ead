"""The stem's backward at the bench geometry: kodhip_stem_bwd_fused against the two launches it replaces
(kodhip_bn_silu_bwd_apply + kodhip_conv_wgrad in the stem form)."""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from object_detection_cib_amd import _lib
from hip_helpers import stream
lib = _lib.lib()
B, H, W, N = 64, 640, 640, 32
Wp, M = W // 2, B * (H // 2) * (W // 2)
x = torch.rand(B, H, Wp, 8, device="cuda").to(torch.bfloat16)
y = torch.randn(M, N, device="cuda").to(torch.bfloat16)
dA = torch.randn(M, N, device="cuda").to(torch.bfloat16)
scale, shift = torch.rand(N, device="cuda") + 0.5, torch.randn(N, device="cuda") * 0.3
coef = torch.cat([torch.rand(N) + 0.5, torch.randn(N) * 0.05, torch.randn(N) * 0.05]).cuda()
blocks = lib.kodhip_stem_bwd_fused_blocks(B, H, Wp, N)
part = torch.empty(blocks * 32 * 160, device="cuda")
gw = torch.zeros(N, 3, 6, 6, device="cuda")
splits = lib.kodhip_conv_wgrad_splits_geo(B, H, Wp, 8, 8, N, 6, 3, 2, 1, 2, 1, 160, N)
part2 = torch.empty(splits * N * 160, device="cuda")
dy = y.clone()


def fused():
    _lib.check(lib.kodhip_stem_bwd_fused(x.data_ptr(), dA.data_ptr(), N, 0, y.data_ptr(), N, scale.data_ptr(), shift.data_ptr(),
                                         coef.data_ptr(), part.data_ptr(), gw.data_ptr(), B, H, Wp, N, 1.0, stream()))


def apply():
    _lib.check(lib.kodhip_bn_silu_bwd_apply(dA.data_ptr(), N, 0, dy.data_ptr(), N, scale.data_ptr(), shift.data_ptr(),
                                            coef.data_ptr(), None, 0, 0, 0, M, N, stream()))


def wgrad():
    _lib.check(lib.kodhip_conv_wgrad(x.data_ptr(), dy.data_ptr(), part2.data_ptr(), gw.data_ptr(), B, H, Wp, 8, 0, 8, N, 6, 3,
                                     2, 1, 2, 1, 160, N, 0, N, 1, 1.0, stream()))


def t(f, name, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    print("%-28s %.1f us" % (name, e0.elapsed_time(e1) * 1e3 / n))


t(fused, "stem bwd fused (+reduce)")
t(apply, "bn_silu_bwd_apply")
t(wgrad, "stem wgrad (+reduce)")
