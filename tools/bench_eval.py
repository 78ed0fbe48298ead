"""Validation loop throughput: device resize/letter-box -> eval forward -> decode -> NMS -> mAP matching.  Diagnostic."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import synth
import bench
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
rep = exp.validate(list(batches(1)), nc)
torch.cuda.synchronize()
n = 6
if "--profile" in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
rep = exp.validate(list(batches(n)), nc)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
if "--profile" in sys.argv:
    pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(22)
print(f"validation: {dt*1e3:.1f} ms/batch = {B/dt:.0f} img/s (random-init weights: worst case box counts); keys {list(rep)[:4]}")
