"""Validation loop throughput: device resize/letter-box -> eval forward -> decode -> NMS -> mAP matching.  Diagnostic."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import synth
import bench
from object_detection_cib_amd import _lib as _kl; _kl.limit_host_threads()
from object_detection_cib_amd.core.anchors.info import voc_anchor_info
from object_detection_cib_amd.data.device_pipeline import DeviceValPipeline
from object_detection_cib_amd.lightning.experiments.yv5_baseline.exp import DefaultYolov5Experiment
from object_detection_cib_amd.lightning.experiments.yv5_baseline.type_defs import LayerwiseAnchorInfo

B, S, nc = 64, 640, 10
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
cache = synth.coco_zipf_like(256, 500, 3, nc)          # original-size images (longest side 500) -> resized to 640
pipe = DeviceValPipeline([c[0] for c in cache], [c[1] for c in cache], [c[2] for c in cache], S, dev)
net, loss = bench.build(nc, dev)
infos = (voc_anchor_info(8), voc_anchor_info(16), voc_anchor_info(32))
exp = DefaultYolov5Experiment(net, loss, LayerwiseAnchorInfo(*infos), graphed="--eager" not in sys.argv)
# a few training-mode forwards so that BN running stats are sane, then eval
x, tg = bench.synth_batch(B, S, nc, 1, dev)
for _ in range(2): net(x)
torch.cuda.synchronize()
def batches(n):
    for i in range(n):
        img, _, t = pipe.make_batch([(i * B + k) % 256 for k in range(B)])
        yield (img, t, None)
rep = exp.validate(batches(8), nc)     # warm-up: allocator pools, pinned staging rings, graph capture
torch.cuda.synchronize()
n = 16
if "--profile" in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
rep = exp.validate(batches(n), nc)            # batches are produced one at a time, like a validation loop does
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
if "--profile" in sys.argv:
    pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(22)
if "--stages" in sys.argv:          # the same loop with a synchronize + clock around every stage
    from object_detection_cib_amd.lightning.callbacks.map_eval import DeviceMAPEvaluator
    ev = DeviceMAPEvaluator(nc)
    def T(fn):
        torch.cuda.synchronize(); a = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return r, (time.perf_counter() - a) * 1e3
    it = iter(batches(12))
    for i in range(12):
        b, t_prep = T(lambda: next(it))
        (tg, dets), t_val = T(lambda: exp.validation_step(b))
        _, t_map = T(lambda: ev.add_batch(tg, dets))
        ms = torch.cuda.memory_stats()
        print(f"stage times batch {i}: prep {t_prep:.2f} validation_step {t_val:.2f} add_batch {t_map:.2f} ms | device allocs "
              f"{ms.get('num_device_alloc')} frees {ms.get('num_device_free')} reserved {ms.get('reserved_bytes.all.current', 0) / 1e9:.2f} GB")
print(f"validation: {dt*1e3:.1f} ms/batch = {B/dt:.0f} img/s (random-init weights: worst case box counts); keys {list(rep)[:4]}")
