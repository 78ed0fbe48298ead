"""HIP side of tests/test_hip_training.py::test_first_epoch_map_vs_cpu_trainer as a script: prints the epoch's loss
fifths and mAP (used to sample the run-to-run spread under summation-order switches such as KODHIP_FORCE_BM=128).
KODHIP_FE_ULP=k: element k of the stem's weight starts one bf16 ulp (2^-8 relative) away from the seeded value - the
perturbation oracle/first_epoch.py --extra2 gives the CPU trainer's bf16-storage emulation: another draw of the trajectory."""
import os, random, sys
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from oracle import first_epoch as FE, synth
from test_hip_training import _experiment
from object_detection_cib_amd.data.device_pipeline import DeviceTrainPipeline
from object_detection_cib_amd.data.detection import DetectionTarget

cfg = FE.CONFIG
S, B, nc, seed = cfg["S"], cfg["B"], cfg["nc"], cfg["seed"]
train = synth.coco_zipf_like(cfg["n_train"], S, cfg["data_seed"], nc)
val = synth.coco_zipf_like(cfg["n_val"], S, cfg["data_seed"] + 1, nc)
pipe = DeviceTrainPipeline([c[0] for c in train], [c[1] for c in train], [c[2] for c in train], S, "cuda", rng_seed=51)
exp = _experiment(cfg["widen"], cfg["deepen"], nc, seed)
exp.val_nms_conf_threshold, exp.val_nms_iou_threshold = cfg["conf_thres"], cfg["nms_thres"]
ulp = int(os.environ.get("KODHIP_FE_ULP", "-1"))
if ulp >= 0:
    with torch.no_grad():
        w = next(exp.net.parameters()).view(-1)
        w[ulp] = w[ulp] * (1.0 + 2.0 ** -8)
    exp.net.engine().mark_params_changed()
order = FE.epoch_order(cfg)
n_batches = len(order) // B
random.seed(seed); np.random.seed(seed)
losses = []
for step in range(n_batches):
    img, _, targets = pipe.make_batch([int(i) for i in order[step * B:(step + 1) * B]])
    losses.append(exp.optimize((img, targets, None), n_batches).detach())
exp.end_epoch()
hip = torch.stack(losses).cpu().numpy()
vb = [(x.cuda(), tuple(DetectionTarget(torch.from_numpy(b), torch.from_numpy(l)) for b, l in tg), None)
      for x, tg in FE.validation_batches(cfg, val)]
rep = exp.validate(vb, nc)
f = n_batches // 5
print("fifths", [round(float(hip[k * f:(k + 1) * f].mean()), 4) for k in range(5)],
      {k: round(v, 4) for k, v in rep.items() if not k.startswith("map50_")}, flush=True)
# cross-check of the evaluation side: the HIP-trained weights evaluated by the CPU oracle's eval pipeline
from oracle.network import OracleYolov5
ref = OracleYolov5(3, nc, cfg["widen"], cfg["deepen"])
ref.load_state_dict({k: v.detach().cpu() for k, v in exp.net.state_dict().items()})
rep_cpu = FE.evaluate(cfg, ref, val)
print("same weights, CPU oracle eval:", {k: round(v, 4) for k, v in rep_cpu.items() if not k.startswith("map50_")}, flush=True)
