"""How many blocks of each fused-forward kernel variant are resident at once IN FACT (the blocks of kodhip_conv_fwd_bn_silu
wait for each other, so a grid beyond that number never finishes its hand-off): launches grids of growing size with a short
poll limit and reports which complete.  KODHIP_FUSE_FORCE=1 switches the library's own capacity check off."""
import os, sys, torch
os.environ["KODHIP_FUSE_FORCE"] = "1"
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from object_detection_cib_amd import _lib
from hip_helpers import pack, stream
lib = _lib.lib()
VARIANTS = {  # name: (Cin, Cout, k, pixels per tile)
    "128x128": (128, 128, 1, 128), "128x64": (128, 64, 1, 128), "256x128": (512, 128, 1, 256), "256x64": (512, 64, 1, 256),
    "row3 128x128": (128, 128, 3, 128), "row3 128x64": (64, 64, 3, 128),
}
for name, (Cin, Cout, k, bm) in VARIANTS.items():
    line = f"{name:14s}"
    for G in (256, 384, 512, 640, 768, 896, 1024):
        H = W = 16
        B = G * bm // (H * W)
        p = k // 2
        nb = lib.kodhip_conv_fwd_bn_silu_ws_bytes(B, H, W, Cin, Cin, Cout, k, k, 1, 1, p, p)
        if nb <= 0:
            line += f" | {G}: -"
            continue
        x = torch.randn(B, H, W, Cin, device="cuda").to(torch.bfloat16)
        pk = pack([torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5])
        raw = torch.empty(B, H, W, Cout, device="cuda", dtype=torch.bfloat16); out = torch.empty_like(raw)
        gamma, beta = torch.ones(Cout, device="cuda"), torch.zeros(Cout, device="cuda")
        rm, rv = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
        aff = torch.zeros(4 * Cout, device="cuda"); err = torch.zeros(4, dtype=torch.int32, device="cuda")
        ws = torch.zeros(nb // 8, dtype=torch.int64, device="cuda")
        _lib.check(lib.kodhip_conv_fwd_bn_silu(x.data_ptr(), pk["f"].data_ptr(), raw.data_ptr(), ws.data_ptr(), B, H, W, Cin, 0, Cin, Cout, k, k, 1, 1, p, p, pk["Kp"], Cout, 0,
                                               gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.03, 1e-3, aff.data_ptr(), 1, None, 0, 0, out.data_ptr(), Cout, 0,
                                               err.data_ptr(), 20000, stream()))
        torch.cuda.synchronize()
        line += f" | {G}: {'ok' if int(err[0]) == 0 else 'STUCK'}"
    print(line, flush=True)
