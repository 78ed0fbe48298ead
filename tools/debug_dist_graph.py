"""Single-GPU rehearsal of the N>1 step under hipGraph capture: per-step losses, eager vs captured, with the
bucket all-reduce in different forms.  usage: KODHIP_FORCE_COLLECTIVES=1 python tools/debug_dist_graph.py MODE
MODE: async (product path) | sync | skip"""
import os, sys
sys.path.insert(0, '.')
os.environ["KODHIP_FORCE_COLLECTIVES"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
import torch, torch.distributed as dist
import bench
from object_detection_cib_amd.engine import executor as ex
from object_detection_cib_amd.core.types import FeatureShape

mode = sys.argv[1] if len(sys.argv) > 1 else "async"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
S = int(sys.argv[3]) if len(sys.argv) > 3 else 320
BUCKET_MB = float(os.environ.get('DBG_BUCKET_MB', '2.0'))
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=0, world_size=1)
if mode == "sync":
    class W:
        def wait(self): pass
    ex.launch_bucket = lambda flat, lo, hi, group=None, stream=None, comm=None: (comm.all_reduce(flat[lo:hi]), W())[1]
elif mode == "skip":
    class W:
        def wait(self): pass
    ex.launch_bucket = lambda flat, lo, hi, group=None, stream=None, comm=None: W()

def run(use_graph, sync_bn):
    net, loss_fn = bench.build(10, dev)
    eng = net.engine()
    net.configure_distributed(None, sync_batchnorm=sync_bn, bucket_mb=BUCKET_MB, native_rccl=True)
    x, targets = bench.synth_batch(B, S, 10, 2023, dev)
    shape = FeatureShape(width=S, height=S)
    eng.sgd_step((0.1, 1e-4, 1e-4), (0.8,) * 3, (0.0, 5e-4, 0.0), 1.0)
    params = list(net.parameters())
    def step():
        for p in params: p.grad = None
        lr_ = loss_fn(shape, net(x), targets)
        total = B * (lr_.localization + lr_.classification + lr_.objectness)
        total.backward(); eng.wait_grads(); eng.sgd_step_device()
        return total
    losses = []
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): losses.append(float(step().item()))
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    if use_graph:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            last = step()
        print("   captured", flush=True)
        for _ in range(5):
            g.replay(); losses.append(float(last.item())); print("   replay", losses[-1], flush=True)
    else:
        for _ in range(5): losses.append(float(step().item()))
    return losses

for sync_bn in (False, True):
    e = run(False, sync_bn); g = run(True, sync_bn)
    print(f"mode={mode} sync_bn={sync_bn}\n  eager {['%.4f' % v for v in e]}\n  graph {['%.4f' % v for v in g]}", flush=True)
dist.destroy_process_group()
