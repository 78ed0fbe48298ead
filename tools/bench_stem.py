"""Stem forward (6x6/s2/p2 on 3 channels as a 6x1-tap wide-pixel conv) at the bench geometry."""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from object_detection_cib_amd import _lib
from hip_helpers import pack, stream
lib = _lib.lib()
B, H, W, N = 64, 640, 640, 32
x = torch.randn(B, H, W // 2, 8, device="cuda").to(torch.bfloat16)
w = torch.randn(N, 3, 6, 6) / 108 ** 0.5
pk = pack([w], stem=True)
y = torch.empty(B, H // 2, W // 2, N, device="cuda", dtype=torch.bfloat16)
T = lib.kodhip_conv_stats_slots(B * (H // 2) * (W // 2), N)
st = torch.empty(2 * N * T, device="cuda")
def f(): _lib.check(lib.kodhip_conv_fwd_raw(x.data_ptr(), pk["f"].data_ptr(), y.data_ptr(), st.data_ptr(), B, H, W // 2, 8, 0, 32, N, 6, 1, 2, 1, 2, 1, pk["Kp"], N, 0, stream()))
for _ in range(3): f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
print("stem fwd %.1f us" % (e0.elapsed_time(e1) * 1e3 / 20))
