"""End-to-end training LOOP throughput: device data pipeline (mosaic/affine/HSV/flip compositing from a u8 pool in
HBM, host-side protocol in Python) -> captured training step.  Diagnostic; bench.py measures the step alone."""
import random, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from oracle import synth
import bench
from object_detection_cib_amd import _lib as _kl; _kl.limit_host_threads()
from object_detection_cib_amd.data.device_pipeline import DeviceTrainPipeline
from object_detection_cib_amd.engine.graphed import GraphedTrainStep

B, S, nc = 64, 640, 10
dev = torch.device("cuda", 0)
cache = synth.coco_zipf_like(256, S, 3, nc)
pipe = DeviceTrainPipeline([c[0] for c in cache], [c[1] for c in cache], [c[2] for c in cache], S, dev)
random.seed(1); np.random.seed(1)
net, loss = bench.build(nc, dev)
net.engine().sgd_step((0.01, 0.01, 0.01), (0.9,) * 3, (0.0, 5e-4, 0.0), 1.0)
t0 = time.perf_counter()
for i in range(5):
    img, _, tg = pipe.make_batch([(i * B + k) % 256 for k in range(B)])
torch.cuda.synchronize()
t_batch = (time.perf_counter() - t0) / 5
gs = GraphedTrainStep(net, loss, B, S, S, max_targets=8192).capture(img, tg)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for i in range(n):
    img, _, tg = pipe.make_batch([(i * B + k) % 256 for k in range(B)])
    total, _ = gs(img, tg)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"make_batch alone {t_batch*1e3:.1f} ms/batch; loop (pipeline + captured step) {dt*1e3:.1f} ms/step = {B/dt:.0f} img/s; loss {float(total):.3f}")
