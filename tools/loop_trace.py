"""The training loop of bench.py's loop leg, alone (for rocprofv3): prints the loop rate; with --const the same batch is
replayed (no host data protocol, no compositing) to separate the step from its feeding."""
import random, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from object_detection_cib_amd import _lib as _kl; _kl.limit_host_threads()
B, S, nc = 64, 640, 10
dev = torch.device("cuda", 0)
net, loss = bench.build(nc, dev)
net.engine().sgd_step((0.01, 0.01, 0.01), (0.9,) * 3, (0.0, 5e-4, 0.0), 1.0)
if "--const" in sys.argv:
    from object_detection_cib_amd.data.device_pipeline import DeviceTrainPipeline
    from object_detection_cib_amd.engine.graphed import GraphedTrainStep
    imgs, boxes, labels = bench.synth_pool(256, S, nc, 7)
    pipe = DeviceTrainPipeline(imgs, boxes, labels, S, dev)
    random.seed(1); np.random.seed(1)
    _, pairs, tg = pipe.make_batch(list(range(B)), out_f32=False, out_pairs=True)
    gs = GraphedTrainStep(net, loss, B, S, S, max_targets=16384, input_pairs=True).capture(pairs, tg)
    for _ in range(3): gs(pairs, tg)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): gs(pairs, tg)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    print(f"const batch: {dt*1e3:.2f} ms/step = {B/dt:.0f} img/s")
else:
    r = bench.loop_leg(net, loss, B, S, nc, dev, 30)
    print(r["ms_per_step"], r["value"])
