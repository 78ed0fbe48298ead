"""Where the training LOOP's time goes beyond the replayed step (bench.py loop leg): the same captured step timed with the
loop's per-step extras added one at a time.  usage (GPU box): python tools/loop_parts.py"""
import random, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from bench import build, synth_pool
from object_detection_cib_amd.data.device_pipeline import DeviceTrainPipeline
from object_detection_cib_amd.engine.graphed import GraphedTrainStep

B, S, nc, N = 64, 640, 10, 30
from object_detection_cib_amd import _lib
_lib.limit_host_threads()          # (torch's host pool sized to the cgroup's CPU share, as bench.py's loop leg does)
dev = torch.device("cuda", 0)
net, loss_fn = build(nc, dev)
net.engine().sgd_step((0.1, 1e-4, 1e-4), (0.8,) * 3, (0.0, 5e-4, 0.0), 1.0)
imgs, boxes, labels = synth_pool(256, S, nc, 7)
pipe = DeviceTrainPipeline(imgs, boxes, labels, S, dev)
random.seed(1); np.random.seed(1)
batches = []
for i in range(4):
    d, m, per = pipe.host.batch([(i * B + k) % 256 for k in range(B)])
    from object_detection_cib_amd.data.host_protocol import pack_targets
    batches.append((d, m, pack_targets(per)))
_, pairs0 = pipe.compose_host_batch(batches[0][0], batches[0][1], out_f32=False, out_pairs=True)
gs = GraphedTrainStep(net, loss_fn, B, S, S, max_targets=16384, input_pairs=True).capture(pairs0, batches[0][2])
main, prep = torch.cuda.current_stream(), torch.cuda.Stream()
buf = gs.eng.image_buffer(B, S, S)


def timed(name, body):
    for _ in range(3):
        body(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        body(i)
    torch.cuda.synchronize()
    print("%-72s %7.3f ms/step" % (name, 1e3 * (time.perf_counter() - t0) / N), flush=True)


timed("graph.replay() only", lambda i: gs.graph.replay())
timed("+ copy of the batch into the input buffer", lambda i: (buf.copy_(pairs0, non_blocking=True), gs.graph.replay()))
timed("+ target upload (gs._load)", lambda i: (gs._load(pairs0, batches[i % 4][2]), gs.graph.replay()))


def side(i):
    d, m, tg = batches[i % 4]
    prep.wait_stream(main)
    with torch.cuda.stream(prep):
        _, pr = pipe.compose_host_batch(d, m, out_f32=False, out_pairs=True)
        ev = torch.cuda.Event(); ev.record(prep)
    pr.record_stream(main)
    gs._load(pairs0, tg)
    gs.graph.replay()
    main.wait_event(ev)


timed("+ compositing of the next batch on a side stream (the loop)", side)


def inline(i):
    d, m, tg = batches[i % 4]
    _, pr = pipe.compose_host_batch(d, m, out_f32=False, out_pairs=True)
    gs._load(pr, tg)
    gs.graph.replay()


timed("compositing on the MAIN stream in front of the step instead", inline)


def direct(i):
    d, m, tg = batches[i % 4]
    _, pr = pipe.compose_host_batch(d, m, out_f32=False, out_pairs=True, pairs_out=gs.input_buffer())
    gs._load(pr, tg)
    gs.graph.replay()


timed("compositing on the main stream STRAIGHT INTO the input buffer (bench.py's loop)", direct)
timed("compose kernel alone (30 launches)", lambda i: pipe.compose_host_batch(batches[i % 4][0], batches[i % 4][1], out_f32=False, out_pairs=True))
