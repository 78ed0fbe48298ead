import random, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from object_detection_cib_amd import _lib as _kl; _kl.limit_host_threads()
from object_detection_cib_amd.data.device_pipeline import DeviceTrainPipeline
from object_detection_cib_amd.engine.graphed import GraphedTrainStep
B, S, nc = 64, 640, 10
dev = torch.device("cuda", 0)
imgs, boxes, labels = bench.synth_pool(256, S, nc, 7)
pipe = DeviceTrainPipeline(imgs, boxes, labels, S, dev)
random.seed(1); np.random.seed(1)
net, loss = bench.build(nc, dev)
net.engine().sgd_step((0.01, 0.01, 0.01), (0.9,) * 3, (0.0, 5e-4, 0.0), 1.0)
_, pairs, tg = pipe.make_batch(list(range(B)), out_f32=False, out_pairs=True)
gs = GraphedTrainStep(net, loss, B, S, S, max_targets=16384, input_pairs=True).capture(pairs, tg)
torch.cuda.synchronize()
tm = tl = 0.0
n = 30
for i in range(n):
    a = time.perf_counter()
    _, pairs, tg = pipe.make_batch([(i * B + k) % 256 for k in range(B)], out_f32=False, out_pairs=True)
    b = time.perf_counter()
    gs(pairs, tg)
    c = time.perf_counter()
    tm += b - a; tl += c - b
torch.cuda.synchronize()
print(f"host: make_batch {tm/n*1e3:.2f} ms, graphed step call {tl/n*1e3:.2f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(10):
    _, pairs, tg = pipe.make_batch([(i * B + k) % 256 for k in range(B)], out_f32=False, out_pairs=True)
    gs(pairs, tg)
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(14)
