"""Throughput of one training step for another network scale (e.g. yv5m: 0.75 0.67) - diagnostic, not the bench."""
import sys, time
sys.path.insert(0, '.')
import torch
import bench
from object_detection_cib_amd.core.anchors.info import voc_anchor_info
from object_detection_cib_amd.core.bbox.iou import IoUCalculator
from object_detection_cib_amd.core.label_assignment.yv5 import Yolov5LabelAssigner, AssignmentAnchorInfo
from object_detection_cib_amd.core.types import FeatureShape
from object_detection_cib_amd.lightning.experiments.yv5_baseline.loss import Yolov5Loss, Yolov5LossParams
from object_detection_cib_amd.nn.networks.yolov5 import Yolov5Network

widen, deepen = float(sys.argv[1]), float(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
S, nc = 640, 10
dev = torch.device("cuda", 0)
torch.manual_seed(2023)
net = Yolov5Network(3, nc, widen_factor=widen, deepen_factor=deepen).to(dev).train()
asg = Yolov5LabelAssigner(AssignmentAnchorInfo(voc_anchor_info(8), voc_anchor_info(16), voc_anchor_info(32)), 4.0)
loss_fn = Yolov5Loss(asg, Yolov5LossParams.get_default(), IoUCalculator("ciou", 1e-7), None)
eng = net.engine()
x, targets = bench.synth_batch(B, S, nc, 2023, dev)
shape = FeatureShape(width=S, height=S)
eng.sgd_step((0.1, 1e-4, 1e-4), (0.8,) * 3, (0.0, 5e-4, 0.0), 1.0)
params = list(net.parameters())

def step():
    for p in params: p.grad = None
    lr_ = loss_fn(shape, net(x), targets)
    total = B * (lr_.localization + lr_.classification + lr_.objectness)
    total.backward(); eng.wait_grads(); eng.sgd_step_device()
    return total

side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): last = step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    last = step()
g.replay(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): g.replay()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"widen {widen} deepen {deepen} B={B}: {dt*1e3:.2f} ms/step, {B/dt:.0f} img/s, loss {float(last.item()):.4f}, params {sum(p.numel() for p in params)}")
