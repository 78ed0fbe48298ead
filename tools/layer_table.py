"""Per-unit timing table of one eager training step (bench.py's event-timed profile, one stream, families not overlapping):
for every conv unit its forward conv / statistics finalize / apply, BatchNorm-backward coefficients / apply, data gradient,
weight gradient (+ slab reduction), with the algorithmic GB/s and TF/s of the three convolution passes and the share of the
step spent per stride level.  usage (GPU box): python tools/layer_table.py [yv5s|yv5m] > profiles/<tag>_layer_table.txt"""
import sys
import torch
sys.path.insert(0, ".")
from bench import build, synth_batch, VARIANTS, profile_step
from object_detection_cib_amd.core.types import FeatureShape

variant = sys.argv[1] if len(sys.argv) > 1 else "yv5s"
B, S, nc = 64, 640, 10
dev = torch.device("cuda", 0)
widen, deepen = VARIANTS[variant]
net, loss_fn = build(nc, dev, widen=widen, deepen=deepen)
eng = net.engine()
x, targets = synth_batch(B, S, nc, 2023, dev)
shape = FeatureShape(width=S, height=S)
eng.sgd_step((0.1, 1e-4, 1e-4), (0.8,) * 3, (0.0, 5e-4, 0.0), 1.0)
params = list(net.parameters())


def step():
    for p in params:
        p.grad = None
    net.train_step(x, loss_fn, shape, targets, float(B))
    eng.wait_grads()
    eng.sgd_step_device()


prof = profile_step(eng, step)
per = {}                       # unit name -> {family: us}
for fam, e0, e1, nb, name in prof:
    us = 1e3 * e0.elapsed_time(e1)
    for n in (name.split("+") if name else ["?"]):
        # a launch shared by two units (dual data / weight gradient, pair coefficients) is booked to the first, flagged on the second
        per.setdefault(n, {})
    first = name.split("+")[0] if name else "?"
    per[first][fam] = per[first].get(fam, 0.0) + us
    for other in name.split("+")[1:]:
        per[other].setdefault("shared_with", first)
units = [op.unit for op in eng.g.ops if op.kind == "conv"]
print(f"# {variant}, B={B}, {S} px, one eager step on one stream (event-timed; the weight gradients' time includes their slab reductions)")
print("# us per launch; GB/s = algorithmic bytes 2*(M_in*Cin + M*Cout) / time, TF = 2*M*Cout*Cin*k*k / time; '=' the launch is shared with the unit named at the end")
print("%-46s %4s %4s %2s %8s | %6s %5s %4s | %5s %6s | %6s %6s | %6s %5s %4s | %6s %5s %4s" % (
    "unit", "cin", "cout", "ks", "M", "fwd", "GB/s", "TF", "fin", "apply", "coeff", "bapply", "dgrad", "GB/s", "TF", "wgrad", "GB/s", "TF"))
by_stride, tot = {}, 0.0
for u in units:
    st = eng.ustate[u.name]
    cin = 3 if u.stem else u.cin
    k = 6 if u.stem else u.k
    Min = B * (S // (1 if u.stem else u.src.stride)) ** 2
    byts, fl = 2.0 * (Min * cin + st.M * u.cout), 2.0 * st.M * u.cout * cin * k * k
    d = per.get(u.name, {})
    g = lambda f: d.get(f, 0.0)
    dg = g("dgrad") + g("dgrad+bn_reduce")
    cell = lambda us: ("%6.1f %5.0f %4.0f" % (us, byts / us / 1e3, fl / us / 1e6)) if us > 0 else ("%6s %5s %4s" % ("=", "", ""))
    row_us = sum(v for kk, v in d.items() if kk != "shared_with")
    stride = u.dst.stride
    by_stride[stride] = by_stride.get(stride, 0.0) + row_us
    tot += row_us
    print("%-46s %4d %4d %d%d %8d | %s | %5.1f %6.1f | %6.1f %6.1f | %s | %s%s" % (
        u.name[-46:], cin, u.cout, k, u.s, st.M, cell(g("conv_fwd")), g("bn_finalize"), g("bn_silu_apply"),
        g("bn_bwd_coeffs") + g("bn_bwd_reduce"), g("bn_silu_bwd_apply"), cell(dg), cell(g("wgrad")),
        ("   = " + d["shared_with"][-30:]) if "shared_with" in d else ""))
other = sum(1e3 * e0.elapsed_time(e1) for fam, e0, e1, nb, name in prof if not name or name.split("+")[0] not in {u.name for u in units})
print("# time per output stride of the units (us; conv units only, heads / pools / loss / optimizer = %.0f us beside them):" % other)
for stv in sorted(by_stride):
    print("#   stride %2d: %7.0f us  %4.1f %%" % (stv, by_stride[stv], 100.0 * by_stride[stv] / tot))
print("#   total    : %7.0f us" % tot)
