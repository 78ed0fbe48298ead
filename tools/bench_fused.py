"""conv + BatchNorm statistics + apply: ONE launch (kodhip_conv_fwd_bn_silu) against the three-launch form, per layer
(yv5s, B=64, strides 16 / 32), back to back on one stream: us per unit."""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from object_detection_cib_amd import _lib
from hip_helpers import pack, stream

lib = _lib.lib()
LAYERS = {  # name: (Cin, H, W, Cout, k, s, p)
    "s4.main 512->256 1x1 @20": (512, 20, 20, 256, 1, 1, 0),
    "s4.conv1 256->256 1x1 @20": (256, 20, 20, 256, 1, 1, 0),
    "s4.b.conv2 256->256 3x3 @20": (256, 20, 20, 256, 3, 1, 1),
    "s4.last 512->512 1x1 @20": (512, 20, 20, 512, 1, 1, 0),
    "sppf.conv2 1024->512 1x1 @20": (1024, 20, 20, 512, 1, 1, 0),
    "s4.conv 256->512 3x3s2 @40": (256, 40, 40, 512, 3, 2, 1),
    "s3.main 256->128 1x1 @40": (256, 40, 40, 128, 1, 1, 0),
    "s3.conv1 128->128 1x1 @40": (128, 40, 40, 128, 1, 1, 0),
    "s3.b.conv2 128->128 3x3 @40": (128, 40, 40, 128, 3, 1, 1),
    "td0.main 512->128 1x1 @40": (512, 40, 40, 128, 1, 1, 0),
    "s3.last 256->256 1x1 @40": (256, 40, 40, 256, 1, 1, 0),
    "ds0 128->128 3x3s2 @80": (128, 80, 80, 128, 3, 2, 1),
}
B = 64
for name in ((sys.argv[1:] or list(LAYERS)) if __name__ == "__main__" else []):
    Cin, H, W, Cout, k, s, p = LAYERS[name]
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    M = B * Ho * Wo
    x = torch.randn(B, H, W, Cin, device="cuda").to(torch.bfloat16)
    pk = pack([torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5])
    raw = torch.empty(B, Ho, Wo, Cout, device="cuda", dtype=torch.bfloat16)
    out = torch.empty_like(raw)
    T = lib.kodhip_conv_stats_slots(M, Cout)
    stats = torch.empty(2 * Cout * T, device="cuda")
    gamma, beta = torch.ones(Cout, device="cuda"), torch.zeros(Cout, device="cuda")
    rm, rv = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
    aff = torch.zeros(4 * Cout, device="cuda")
    err = torch.zeros(4, dtype=torch.int32, device="cuda")
    nb = lib.kodhip_conv_fwd_bn_silu_ws_bytes(B, H, W, Cin, Cin, Cout, k, k, s, s, p, p)
    ws = torch.zeros(max(nb // 8, 1), dtype=torch.int64, device="cuda")
    a = aff.data_ptr()

    def three():
        _lib.check(lib.kodhip_conv_fwd_raw(x.data_ptr(), pk["f"].data_ptr(), raw.data_ptr(), stats.data_ptr(), B, H, W, Cin, 0, Cin, Cout, k, k, s, s, p, p, pk["Kp"], Cout, 0, stream()))
        _lib.check(lib.kodhip_bn_finalize_partials(stats.data_ptr(), T, float(M), gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.03, 1e-3, a, a + 4 * Cout, a + 8 * Cout, a + 12 * Cout, Cout, 1, stream()))
        _lib.check(lib.kodhip_bn_silu_apply(raw.data_ptr(), Cout, a, a + 4 * Cout, None, 0, 0, out.data_ptr(), Cout, 0, M, Cout, stream()))

    def one():
        _lib.check(lib.kodhip_conv_fwd_bn_silu(x.data_ptr(), pk["f"].data_ptr(), raw.data_ptr(), ws.data_ptr(), B, H, W, Cin, 0, Cin, Cout, k, k, s, s, p, p, pk["Kp"], Cout, 0,
                                               gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.03, 1e-3, a, 1, None, 0, 0, out.data_ptr(), Cout, 0, err.data_ptr(), 0, stream()))

    line = f"{name:32s}"
    for fn in (three, one) if nb > 0 else (three,):
        for _ in range(3): fn()
        g = torch.cuda.CUDAGraph()                  # replayed like the training step: no host launch cost in the figure
        with torch.cuda.graph(g):
            for _ in range(20): fn()
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(5): g.replay()
        e1.record(); torch.cuda.synchronize()
        line += f" | {fn.__name__:5s} {e0.elapsed_time(e1) * 1e3 / 100:7.1f} us"
    print(line + (f"   (workspace {nb >> 10} KB, err {int(err[0])})" if nb > 0 else "   (not resident: three launches only)"), flush=True)
