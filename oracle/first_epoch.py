"""Oracle: the reference CPU trainer's FIRST EPOCH on a synthetic coco-zipf-like set, and the mAP it reaches.

TEST INFRASTRUCTURE - see oracle/__init__.py.  Restates, with the oracle's pinned pieces, what
`python kod/cli/hydra_train.py experiment=yv5s trainer=cpu ...` does for one epoch:

* per-sample protocol (mosaic + affine + HSV + flip)     kod/data/detection.py:102-156 -> oracle/datapath.py
* training_step / automatic optimisation                  kod/lightning/experiments/yv5_baseline/exp.py:104-138
* warm-up hook + SGD-nesterov groups                      exp.py:164-185, warmup.py:39-58, nn/optim/smart.py:36-58
* validation_step (letter-box, decode, NMS .001 / .6)     exp.py:140-154, kod/core/nms.py:9-75
* mAP report                                              kod/lightning/callbacks/pycoco_map_eval.py:106-125

    python -m oracle.first_epoch            # writes tests/golden/first_epoch.npz (a few minutes of CPU)

The HIP run of the same protocol (tests/test_hip_training.py::test_first_epoch_map_vs_cpu_trainer) feeds identical
batches (the device compositing kernel is bit-exact against oracle/datapath.py) from identical seeds and initial
weights; the two trainers then differ only by bf16 storage vs fp32, which decorrelates the trajectories after a
few hundred steps, so the test compares epoch-level quantities (mean loss per phase, mAP) with the tolerances
stated there.  A second CPU run with bf16-rounded weights/activations (oracle/bf16_emul.py) is recorded as the
reference's own sensitivity to that perturbation.
"""
from __future__ import annotations

import os
import random
import sys
import time

import numpy as np
import torch

from . import datapath, detection as D, map_eval, optim as O, synth
from .network import OracleYolov5

# one config for both trainers; small enough that the CPU side takes minutes, long enough that mAP leaves zero
CONFIG = dict(widen=0.5, deepen=0.33, nc=10, S=160, B=16, n_train=8000, n_val=256, seed=2023, data_seed=77,
              conf_thres=0.001, nms_thres=0.6)


def epoch_order(cfg) -> np.ndarray:
    return np.random.default_rng(cfg["seed"]).permutation(cfg["n_train"])


def validation_batches(cfg, val):
    S, B = cfg["S"], cfg["B"]
    out = []
    for k in range(0, len(val), B):
        imgs, tg = [], []
        for im, bb, lb in val[k:k + B]:
            x, b = datapath.val_sample(im, bb, S)
            imgs.append(torch.from_numpy(np.ascontiguousarray(x)))
            tg.append((b, lb))
        out.append((torch.stack(imgs), tg))
    return out


def evaluate(cfg, net, val):
    net.eval()
    per_image = []
    with torch.no_grad():
        for x, tg in validation_batches(cfg, val):
            det = D.decode(net(x), cfg["S"], cfg["S"])
            for d, (b, l) in zip(D.nms(det, cfg["conf_thres"], cfg["nms_thres"]), tg):
                per_image.append(map_eval.match_image(d.numpy(), b, l, cfg["nc"]))
    net.train()
    return map_eval.report(map_eval.accumulate(per_image, cfg["nc"]))


def run_cpu(cfg=CONFIG, emulate_bf16: bool = False, log=None, ulp: int = -1):
    """ulp >= 0: element `ulp` of the stem's weight starts one ulp away from the seeded value - one fp32 ulp for the fp32
    trainer, one BF16 ulp (2^-8 relative) for the bf16-storage emulation, which rounds the weight to bf16 before it is used
    (an fp32 ulp vanishes there: seven such runs reproduced the unperturbed one digit for digit) - a perturbation of one of
    the network's 7 M weights at the level of the storage rounding itself, enough to give another draw of the chaotic
    first-epoch trajectory (the thread-count trick of main() / extra() has only eight settings on this machine)."""
    S, B, nc, seed = cfg["S"], cfg["B"], cfg["nc"], cfg["seed"]
    train = synth.coco_zipf_like(cfg["n_train"], S, cfg["data_seed"], nc)
    val = synth.coco_zipf_like(cfg["n_val"], S, cfg["data_seed"] + 1, nc)
    torch.manual_seed(seed)
    net = OracleYolov5(3, nc, cfg["widen"], cfg["deepen"]).train()
    if ulp >= 0:
        with torch.no_grad():
            w = next(net.parameters()).view(-1)
            w[ulp] = w[ulp] * (1.0 + 2.0 ** -8) if emulate_bf16 else torch.nextafter(w[ulp], w[ulp] + 1)
    if emulate_bf16:
        from . import bf16_emul
        net = bf16_emul.emulate(net)
    bias, decay, norm = O.param_groups(net)
    opt = torch.optim.SGD([dict(params=bias, weight_decay=0.0), dict(params=decay, weight_decay=O.WEIGHT_DECAY),
                           dict(params=norm, weight_decay=0.0)], lr=O.LR0, momentum=O.MOMENTUM, nesterov=True)
    order = epoch_order(cfg)
    n_batches = len(order) // B
    nw = O.warmup_steps(n_batches)
    random.seed(seed); np.random.seed(seed)
    rng = np.random.default_rng(51)
    losses = np.zeros((n_batches, 4))
    t0 = time.time()
    for step in range(n_batches):
        samples = [datapath.train_sample(train, int(i), S, rng) for i in order[step * B:(step + 1) * B]]
        x = torch.from_numpy(np.stack([s[0] for s in samples]))
        tg = [D.Target(torch.from_numpy(s[1]), torch.from_numpy(s[2])) for s in samples]
        w = O.warmup_values(step, 0, nw)
        for pg, name in zip(opt.param_groups, O.GROUP_NAMES):
            pg["lr"], pg["momentum"] = w[name]
        opt.zero_grad(set_to_none=True)
        lr = D.yolo_loss(S, S, net(x), tg)
        tot = D.train_step_total(lr, B)
        tot.backward()
        opt.step()
        losses[step] = (lr.localization.item(), lr.objectness.item(), lr.classification.item(), tot.item())
        if log and (step % 50 == 0 or step == n_batches - 1):
            log(f"step {step}/{n_batches} total {tot.item():.4f}  ({time.time() - t0:.0f}s)")
    rep = evaluate(cfg, net, val)
    return dict(losses=losses, report=rep)


def main():
    """CPU runs of the same epoch: fp32 on all cores (THE trajectory fixture), fp32 on 4 / 6 / 3 / 5 threads (same
    arithmetic, another summation order inside torch's kernels: the trainer's own run-to-run spread after ~500 chaotic
    steps - first-epoch mAP is a noisy statistic), and fp32 with bf16-rounded storage (the perturbation the HIP path
    applies).  About half an hour on 8 cores."""
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "first_epoch.npz")
    log = lambda s: print(s, file=sys.stderr, flush=True)
    ncpu = os.cpu_count() or 1
    keys = ("map", "map30", "map50", "map75", "map90")
    res, samples = {}, []
    for tag, emu, threads in (("fp32", False, ncpu), ("fp32_alt", False, max(1, ncpu // 2)), ("bf16emu", True, ncpu),
                              ("fp32_t6", False, 6), ("fp32_t3", False, 3), ("fp32_t5", False, 5)):
        torch.set_num_threads(threads)
        r = run_cpu(CONFIG, emu, log)
        log(f"{tag} ({threads} threads): { {k: round(r['report'][k], 4) for k in keys} }")
        if tag in ("fp32", "fp32_alt", "bf16emu"):
            res[f"losses_{tag}"] = r["losses"]
        res[f"map_{tag}"] = np.array([r["report"][k] for k in keys])
        samples.append(res[f"map_{tag}"])
    res["map_cpu_samples"] = np.stack(samples)           # [6 runs, 5 metrics]: five fp32 summation orders + the bf16 emulation
    res["map_sample_tags"] = np.array(["fp32", "fp32_alt", "bf16emu", "fp32_t6", "fp32_t3", "fp32_t5"])      # (--extra appends)
    np.savez_compressed(out, config=np.array([repr(sorted(CONFIG.items()))]), map_keys=np.array(keys), **res)
    print(f"wrote {out}")


def extra():
    """`python -m oracle.first_epoch --extra`: more samples of the CPU trainer's run-to-run spread, appended to the
    fixture (round 3: the HIP-vs-CPU mAP comparison wants a tighter estimate of the spread, and the bf16-storage
    emulation - the HIP path's own perturbation - sampled more than once).  Thread counts not used by main()."""
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "first_epoch.npz")
    log = lambda s: print(s, file=sys.stderr, flush=True)
    keys = ("map", "map30", "map50", "map75", "map90")
    res = dict(np.load(out))
    tags = [str(t) for t in res.get("map_sample_tags", np.array(["fp32", "fp32_alt", "bf16emu", "fp32_t6", "fp32_t3", "fp32_t5"]))]
    samples = [row for row in res["map_cpu_samples"]]
    for tag, emu, threads in (("bf16emu_t4", True, 4), ("fp32_t7", False, 7), ("bf16emu_t6", True, 6), ("fp32_t2", False, 2),
                              ("bf16emu_t3", True, 3), ("bf16emu_t5", True, 5)):
        if tag in tags:
            continue
        torch.set_num_threads(threads)
        r = run_cpu(CONFIG, emu, log)
        log(f"{tag} ({threads} threads): { {k: round(r['report'][k], 4) for k in keys} }")
        tags.append(tag)
        samples.append(np.array([r["report"][k] for k in keys]))
        res["map_cpu_samples"] = np.stack(samples)
        res["map_sample_tags"] = np.array(tags)
        np.savez_compressed(out, **res)               # after every run: a stopped job keeps what it has
    print(f"wrote {out}: {len(samples)} samples")


def extra2():
    """`python -m oracle.first_epoch --extra2`: round 4 - more draws of BOTH CPU trainers (bf16-storage emulation first: the
    HIP-vs-CPU comparison that matters rested on five of them), each from initial weights one fp32 ulp away in one element."""
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "first_epoch.npz")
    log = lambda s: print(s, file=sys.stderr, flush=True)
    keys = ("map", "map30", "map50", "map75", "map90")
    res = dict(np.load(out))
    tags = [str(t) for t in res["map_sample_tags"]]
    samples = [row for row in res["map_cpu_samples"]]
    torch.set_num_threads(os.cpu_count() or 1)
    plan = [(f"bf16emu_u{k}", True, k) for k in range(1, 8)] + [(f"fp32_u{k}", False, k) for k in range(1, 6)]
    for tag, emu, k in plan:
        if tag in tags:
            continue
        r = run_cpu(CONFIG, emu, log, ulp=k)
        log(f"{tag}: { {kk: round(r['report'][kk], 4) for kk in keys} }")
        tags.append(tag)
        samples.append(np.array([r["report"][kk] for kk in keys]))
        res["map_cpu_samples"] = np.stack(samples)
        res["map_sample_tags"] = np.array(tags)
        np.savez_compressed(out, **res)               # after every run: a stopped job keeps what it has
    print(f"wrote {out}: {len(samples)} samples")


if __name__ == "__main__":
    extra2() if "--extra2" in sys.argv else (extra() if "--extra" in sys.argv else main())
