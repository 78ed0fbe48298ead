"""Where the time of a fused conv + BatchNorm launch goes after its K loop (diagnostic build: tools/build_ablate.sh stamps
-DKOD_FZ_STAMPS, run with KODHIP_LIB=tools/ablate/lib_stamps.so): device clock (us) of the finalizer block and of the last
block of channel tile 0 at: block entry, prologue done (first fetch about to be issued), K loop done, then 0 tail begins, 1 partials published, 2 sums known (designated blocks), 3 constants stored and
drained (finalizer), 4 flag seen, 5 constants in registers, 6 tile stored."""
import sys, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, os.path.join(R, "tests"), os.path.join(R, "tools")]
from object_detection_cib_amd import _lib
from hip_helpers import pack, stream
from bench_fused import LAYERS      # noqa
lib = _lib.lib()
B = 64
for name in (sys.argv[1:] or list(LAYERS)):
    Cin, H, W, Cout, k, s, p = LAYERS[name]
    nb = lib.kodhip_conv_fwd_bn_silu_ws_bytes(B, H, W, Cin, Cin, Cout, k, k, s, s, p, p)
    if nb <= 0:
        continue
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    x = torch.randn(B, H, W, Cin, device="cuda").to(torch.bfloat16)
    pk = pack([torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5])
    raw = torch.empty(B, Ho, Wo, Cout, device="cuda", dtype=torch.bfloat16); out = torch.empty_like(raw)
    gamma, beta = torch.ones(Cout, device="cuda"), torch.zeros(Cout, device="cuda")
    rm, rv = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
    aff = torch.zeros(4 * Cout, device="cuda"); err = torch.zeros(4, dtype=torch.int32, device="cuda")
    ws = torch.zeros(nb // 8, dtype=torch.int64, device="cuda")
    for rep in range(3):
        _lib.check(lib.kodhip_conv_fwd_bn_silu(x.data_ptr(), pk["f"].data_ptr(), raw.data_ptr(), ws.data_ptr(), B, H, W, Cin, 0, Cin, Cout, k, k, s, s, p, p, pk["Kp"], Cout, 0,
                                               gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.03, 1e-3, aff.data_ptr(), 1, None, 0, 0, out.data_ptr(), Cout, 0,
                                               err.data_ptr(), 0, stream()))
        torch.cuda.synchronize()
    w = ws.cpu().tolist()
    f, l = w[15:18] + w[8:15], w[27:30] + w[20:27]
    t0 = min(f[0], l[0])
    fmt = lambda v: " ".join(f"{(t - t0) / 100.0:6.2f}" if t else "     -" for t in v)
    print(f"{name:30s} finalizer: {fmt(f)} | last block: {fmt(l)}   (err {int(err[0])})", flush=True)
