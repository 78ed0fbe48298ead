import random, sys
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from oracle import first_epoch as FE, synth
from test_hip_training import _experiment
from object_detection_cib_amd.data.device_pipeline import DeviceTrainPipeline
from object_detection_cib_amd.data.detection import DetectionTarget
from object_detection_cib_amd.engine.options import EngineOptions
cfg = FE.CONFIG
S, B, nc, seed = cfg["S"], cfg["B"], cfg["nc"], cfg["seed"]
train = synth.coco_zipf_like(cfg["n_train"], S, cfg["data_seed"], nc)
val = synth.coco_zipf_like(cfg["n_val"], S, cfg["data_seed"] + 1, nc)
order = FE.epoch_order(cfg); n_batches = len(order) // B
vb = [(x.cuda(), tuple(DetectionTarget(torch.from_numpy(b), torch.from_numpy(l)) for b, l in tg), None) for x, tg in FE.validation_batches(cfg, val)]
for sw in ({"dual_wgrad": False}, {"stem_bwd_fused": False}, {"dual_wgrad": False, "dual_dgrad": False}):
    pipe = DeviceTrainPipeline([c[0] for c in train], [c[1] for c in train], [c[2] for c in train], S, "cuda", rng_seed=51)
    exp = _experiment(cfg["widen"], cfg["deepen"], nc, seed)
    opts = EngineOptions.from_env()
    for k, v in sw.items(): setattr(opts, k, v)
    exp.net.engine_options = opts
    exp.val_nms_conf_threshold, exp.val_nms_iou_threshold = cfg["conf_thres"], cfg["nms_thres"]
    random.seed(seed); np.random.seed(seed)
    losses = []
    for step in range(n_batches):
        img, _, targets = pipe.make_batch([int(i) for i in order[step * B:(step + 1) * B]])
        losses.append(exp.optimize((img, targets, None), n_batches).detach())
    exp.end_epoch()
    hip = torch.stack(losses).cpu().numpy(); f = n_batches // 5
    rep = exp.validate(vb, nc)
    print(sw, [round(float(hip[k*f:(k+1)*f].mean()),4) for k in range(5)], {k: round(v,4) for k,v in rep.items() if k in ("map","map30","map50")}, flush=True)
    del exp, pipe; torch.cuda.empty_cache()
