"""Short end-to-end training runs: device data path -> HIP network/loss/backward -> fused SGD with the
reference's warm-up schedule -> on-device decode/NMS/mAP.  (1) the loss trajectory tracks the CPU oracle
trained on the very same batches; (2) a from-scratch run learns the synthetic coco-zipf-like set."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import detection as D, optim as O, synth  # noqa: E402
from oracle.network import OracleYolov5  # noqa: E402
from object_detection_cib_amd.core.anchors.info import voc_anchor_info  # noqa: E402
from object_detection_cib_amd.core.bbox.iou import IoUCalculator  # noqa: E402
from object_detection_cib_amd.core.label_assignment.yv5 import Yolov5LabelAssigner, AssignmentAnchorInfo  # noqa: E402
from object_detection_cib_amd.data.device_pipeline import DeviceTrainPipeline  # noqa: E402
from object_detection_cib_amd.lightning.experiments.yv5_baseline.exp import DefaultYolov5Experiment  # noqa: E402
from object_detection_cib_amd.lightning.experiments.yv5_baseline.loss import Yolov5Loss, Yolov5LossParams  # noqa: E402
from object_detection_cib_amd.lightning.experiments.yv5_baseline.type_defs import LayerwiseAnchorInfo  # noqa: E402
from object_detection_cib_amd.lightning.experiments.yv5_baseline.warmup import OptimizerWarmupUpdater  # noqa: E402
from object_detection_cib_amd.nn.networks.yolov5 import Yolov5Network  # noqa: E402


def _experiment(widen, deepen, nc, seed):
    torch.manual_seed(seed)
    net = Yolov5Network(3, nc, widen_factor=widen, deepen_factor=deepen).cuda().train()
    infos = (voc_anchor_info(8), voc_anchor_info(16), voc_anchor_info(32))
    loss = Yolov5Loss(Yolov5LabelAssigner(AssignmentAnchorInfo(*infos), 4.0), Yolov5LossParams.get_default(),
                      IoUCalculator("ciou", 1e-7), None)
    return DefaultYolov5Experiment(net, loss, LayerwiseAnchorInfo(*infos),
                                   optimizer_warmup_updater=OptimizerWarmupUpdater(3, 0.1, 0.8, 0.937))


def test_loss_trajectory_tracks_cpu_oracle():
    S, nc, B, steps, seed = 160, 10, 8, 30, 4
    cache = synth.coco_zipf_like(64, S, seed, nc)
    pipe = DeviceTrainPipeline([c[0] for c in cache], [c[1] for c in cache], [c[2] for c in cache], S, "cuda")
    random.seed(seed); np.random.seed(seed)
    exp = _experiment(0.25, 0.33, nc, seed)
    torch.manual_seed(seed)
    ref = OracleYolov5(3, nc, 0.25, 0.33).train()
    bias, decay, norm = O.param_groups(ref)
    opt = torch.optim.SGD([dict(params=bias, weight_decay=0.0), dict(params=decay, weight_decay=5e-4),
                           dict(params=norm, weight_decay=0.0)], lr=0.01, momentum=0.937, nesterov=True)
    hip, cpu = [], []
    n_batches = 8
    nw = max(round(n_batches * 3), 100)
    for step in range(steps):
        idx = [(step * B + k) % len(cache) for k in range(B)]
        img, _, targets = pipe.make_batch(idx)
        hip.append(exp.optimize((img, targets, None), n_batches).item())
        w = O.warmup_values(step, 0, nw)
        for pg, name in zip(opt.param_groups, O.GROUP_NAMES):
            pg["lr"], pg["momentum"] = w[name]
        opt.zero_grad()
        tot = D.train_step_total(D.yolo_loss(S, S, ref(img.cpu()), [D.Target(t.boxes, t.labels) for t in targets]), B)
        tot.backward()
        opt.step()
        cpu.append(tot.item())
    hip, cpu = np.array(hip), np.array(cpu)
    assert np.isfinite(hip).all() and np.isfinite(cpu).all()
    rel = np.abs(hip - cpu) / np.abs(cpu)
    # bf16 storage vs fp32: a few 1e-3 per step at start, drifting apart slowly as the two runs decorrelate
    assert rel[:5].max() < 1e-2 and rel.max() < 5e-2, rel
    assert hip[-5:].mean() < hip[:5].mean()


def test_training_learns_synthetic_set():
    S, nc, B, seed = 160, 10, 16, 11
    train = synth.coco_zipf_like(256, S, seed, nc)
    val = synth.coco_zipf_like(48, S, seed + 1, nc)
    pipe = DeviceTrainPipeline([c[0] for c in train], [c[1] for c in train], [c[2] for c in train], S, "cuda")
    random.seed(seed); np.random.seed(seed)
    exp = _experiment(0.25, 0.33, nc, seed)
    n_batches = len(train) // B
    first = last = None
    for epoch in range(20):
        order = np.random.permutation(len(train))
        batches = []
        for k in range(n_batches):
            img, _, t = pipe.make_batch(order[k * B:(k + 1) * B].tolist())
            batches.append((img, t, None))
        losses = exp.fit_epoch(batches)
        first = losses.mean().item() if first is None else first
        last = losses.mean().item()
    assert last < 0.8 * first, (first, last)
    # validation: letter-boxed S x S images (pad with 114), /255, CHW  (detection.py:130-132, albu.py:91-119)
    from object_detection_cib_amd.data.detection import DetectionTarget
    vb = []
    for k in range(0, len(val), B):
        imgs, tg = [], []
        for im, bb, lb in val[k:k + B]:
            canvas = np.full((S, S, 3), 114, dtype=np.uint8)
            h, w = im.shape[:2]
            y0, x0 = (S - h) // 2, (S - w) // 2
            canvas[y0:y0 + h, x0:x0 + w] = im
            imgs.append(torch.from_numpy(canvas).permute(2, 0, 1).float() / 255)
            tg.append(DetectionTarget(torch.from_numpy(bb + np.array([x0, y0, x0, y0])), torch.from_numpy(lb)))
        vb.append((torch.stack(imgs).cuda(), tuple(tg), None))
    rep = exp.validate(vb, nc)
    assert set(rep) >= {"map", "map30", "map50", "map75", "map90"} and np.isfinite(rep["map"])
    assert rep["map30"] > 0.01, rep          # learned something real in 320 steps (100 of them warm-up)
    print("synthetic-set report:", {k: round(v, 4) for k, v in rep.items() if not k.startswith("map50_")}, first, last)


def test_graphed_training_loop_equals_eager_loop():
    """DefaultYolov5Experiment(graphed=True) replays one captured hipGraph per step (engine/graphed.py): same batches,
    same warm-up schedule => the same losses and parameters as the eager loop, bit for bit (targets are padded to
    a fixed capacity with zero-size boxes, which the assigner never matches; capture restores the state it touched)."""
    S, nc, B, steps, seed = 160, 10, 8, 8, 4          # (as above: every level gets matches, no NaN level)
    cache = synth.coco_zipf_like(64, S, seed, nc)
    runs = {}
    for graphed in (False, True):
        pipe = DeviceTrainPipeline([c[0] for c in cache], [c[1] for c in cache], [c[2] for c in cache], S, "cuda")
        random.seed(seed); np.random.seed(seed)
        exp = _experiment(0.25, 0.33, nc, seed)
        exp.graphed, exp.max_targets = graphed, 512
        losses = []
        for step in range(steps):
            idx = [(step * B + k) % len(cache) for k in range(B)]
            img, _, targets = pipe.make_batch(idx)
            losses.append(exp.optimize((img, targets, None), 8).item())
        torch.cuda.synchronize()
        runs[graphed] = (losses, torch.cat([p.detach().flatten() for p in exp.net.parameters()]).cpu(),
                         exp.net.engine().rm_arena.cpu().clone())
    assert np.isfinite(runs[False][0]).all()
    assert runs[True][0] == runs[False][0], (runs[True][0], runs[False][0])
    assert torch.equal(runs[True][1], runs[False][1])
    assert torch.equal(runs[True][2], runs[False][2])


def test_graphed_step_from_pixel_pairs_equals_fp32_input():
    """GraphedTrainStep(input_pairs=True): the batch arrives as bf16 pixel pairs (DeviceTrainPipeline.make_batch(out_pairs=
    True), the layout the first layer reads) and goes straight into the network's input buffer - no fp32 NCHW batch, no
    layout-change pass in the captured step; the next batch composited on a side stream while the step runs (bench.py's
    loop leg).  Same losses and parameters, bit for bit, as the fp32-input step on the same batches."""
    from object_detection_cib_amd.engine.graphed import GraphedTrainStep
    from bench import build
    S, nc, B, steps, seed = 160, 10, 8, 5, 6
    cache = synth.coco_zipf_like(64, S, seed, nc)
    runs = {}
    for pairs_mode in (False, True):
        pipe = DeviceTrainPipeline([c[0] for c in cache], [c[1] for c in cache], [c[2] for c in cache], S, "cuda")
        random.seed(seed); np.random.seed(seed)
        torch.manual_seed(seed)
        net, loss_fn = build(nc, torch.device("cuda", 0), seed=seed, widen=0.25)
        net.engine().sgd_step((0.01, 0.01, 0.01), (0.9,) * 3, (0.0, 5e-4, 0.0), 1.0)
        main, prep = torch.cuda.current_stream(), torch.cuda.Stream()

        def produce(i):
            idx = [(i * B + k) % len(cache) for k in range(B)]
            if not pairs_mode:
                img, _, tg = pipe.make_batch(idx)
                return img, tg, None
            prep.wait_stream(main)
            with torch.cuda.stream(prep):
                _, pr, tg = pipe.make_batch(idx, out_f32=False, out_pairs=True)
                ev = torch.cuda.Event()
                ev.record(prep)
            pr.record_stream(main)
            return pr, tg, ev
        first = produce(0)
        if first[2] is not None:
            main.wait_event(first[2])
        gs = GraphedTrainStep(net, loss_fn, B, S, S, max_targets=512, input_pairs=pairs_mode).capture(first[0], first[1])
        losses, nxt = [], first
        for i in range(steps):
            x, tg, ev = nxt
            nxt = produce(i + 1)
            if ev is not None:
                main.wait_event(ev)
            total, _ = gs(x, tg)
            losses.append(float(total))
        torch.cuda.synchronize()
        runs[pairs_mode] = (losses, torch.cat([p.detach().flatten() for p in net.parameters()]).cpu())
    assert np.isfinite(runs[False][0]).all()
    assert runs[True][0] == runs[False][0], (runs[True][0], runs[False][0])
    assert torch.equal(runs[True][1], runs[False][1])


def test_producer_process_loop_equals_in_process_loop():
    """data/producer.py feeding the captured step (bench.py's loop leg; the reference feeds its trainer from DataLoader worker
    processes, kod/lightning/data_module.py:135-144): the host side of the data protocol in a worker process - descriptors,
    mixup ratios and packed targets through shared memory, compositing here - must give the very batches of the in-process
    protocol: images bit for bit, the same losses and parameters after five replayed steps (mixup on: the worker owns
    numpy.random too)."""
    from object_detection_cib_amd.data.producer import DescriptorProducer
    from object_detection_cib_amd.engine.graphed import GraphedTrainStep
    from bench import build
    S, nc, B, steps, seed = 160, 10, 8, 5, 9
    cache = synth.coco_zipf_like(64, S, seed, nc)
    schedule = [[(i * B + k) % len(cache) for k in range(B)] for i in range(steps + 1)]
    runs = {}
    for use_producer in (False, True):
        pipe = DeviceTrainPipeline([c[0] for c in cache], [c[1] for c in cache], [c[2] for c in cache], S, "cuda", mixup_prob=0.4)
        prod = None
        if use_producer:
            prod = DescriptorProducer(pipe.host_args(), B, schedule, rng_seed=51, py_seed=seed, np_seed=seed, max_boxes=512)
        else:
            random.seed(seed); np.random.seed(seed)
        try:
            torch.manual_seed(seed)
            net, loss_fn = build(nc, torch.device("cuda", 0), seed=seed, widen=0.25)
            net.engine().sgd_step((0.01, 0.01, 0.01), (0.9,) * 3, (0.0, 5e-4, 0.0), 1.0)

            def produce(i):
                if prod is not None:
                    descs, mix, tg = prod.next(timeout=120)
                    _, pr = pipe.compose_host_batch(descs, mix, out_f32=False, out_pairs=True)
                    return pr, tg
                _, pr, tg = pipe.make_batch(schedule[i], out_f32=False, out_pairs=True)
                return pr, tg
            first = produce(0)
            gs = GraphedTrainStep(net, loss_fn, B, S, S, max_targets=512, input_pairs=True).capture(first[0], first[1])
            losses, imgs, nxt = [], [first[0].clone()], first
            for i in range(steps):
                x, tg = nxt
                nxt = produce(i + 1)
                imgs.append(nxt[0].clone())
                total, _ = gs(x, tg)
                losses.append(float(total))
            torch.cuda.synchronize()
        finally:
            if prod is not None:
                prod.close()
        runs[use_producer] = (losses, torch.cat([p.detach().flatten() for p in net.parameters()]).cpu(), [t.cpu() for t in imgs])
    assert np.isfinite(runs[False][0]).all()
    for a, b in zip(runs[True][2], runs[False][2]):
        assert torch.equal(a, b)
    assert runs[True][0] == runs[False][0], (runs[True][0], runs[False][0])
    assert torch.equal(runs[True][1], runs[False][1])


def test_graphed_loop_survives_validation_at_other_shapes():
    """A captured hipGraph bakes the engine's buffer addresses in.  A validation forward at another batch size /
    resolution between replays must not free or reuse them (Engine.allocate keeps one buffer set per shape):
    graphed fit -> validate(B', S') -> graphed fit == the eager run of the same schedule, bit for bit."""
    S, nc, B, seed = 160, 10, 8, 4
    cache = synth.coco_zipf_like(64, S, seed, nc)
    runs = {}
    for graphed in (False, True):
        pipe = DeviceTrainPipeline([c[0] for c in cache], [c[1] for c in cache], [c[2] for c in cache], S, "cuda")
        random.seed(seed); np.random.seed(seed)
        exp = _experiment(0.25, 0.33, nc, seed)
        exp.graphed, exp.max_targets = graphed, 512
        losses, vals = [], []
        for step in range(9):
            idx = [(step * B + k) % len(cache) for k in range(B)]
            img, _, targets = pipe.make_batch(idx)
            losses.append(exp.optimize((img, targets, None), 8).item())
            if step % 3 == 2:
                # validation at a different batch size and at a different resolution, eagerly, between replays;
                # scribble over freshly allocated memory afterwards so a stale pointer would read garbage
                for vb, vs in ((3, S), (5, 96)):
                    g = torch.Generator().manual_seed(step)
                    vx = torch.rand(vb, 3, vs, vs, generator=g).cuda()
                    _, dets = exp.validation_step((vx, (), None))
                    vals.append(sum(int(d.shape[0]) for d in dets))
                junk = [torch.full((1 << 22,), float("nan"), device="cuda") for _ in range(8)]
                del junk
        torch.cuda.synchronize()
        runs[graphed] = (losses, vals, torch.cat([p.detach().flatten() for p in exp.net.parameters()]).cpu())
    assert np.isfinite(runs[False][0]).all()
    assert runs[True][0] == runs[False][0], (runs[True][0], runs[False][0])
    assert runs[True][1] == runs[False][1]
    assert torch.equal(runs[True][2], runs[False][2])


def test_first_epoch_map_vs_cpu_trainer(golden):
    """north_star: "first-epoch mAP" parity.  The CPU trainer (oracle/first_epoch.py: the reference's per-sample data
    protocol, network, loss, SGD + warm-up, validation, mAP - all fp32 on the host) trained one epoch of the
    synthetic coco-zipf-like set in the build container; its loss trajectory and mAP are the committed fixture
    tests/golden/first_epoch.npz.  The HIP trainer runs the SAME epoch here: same seeds, same initial weights, same
    batches (the device compositing kernel is bit-exact against the CPU protocol), bf16 activation storage.
    The two trajectories decorrelate after a few hundred steps (chaos, not error), so the bars are epoch-level:
      * per-step total loss within 1e-2 rel over the first 5 steps, 5e-2 over the first 50;
      * mean total loss of each fifth of the epoch: 2e-2 rel for the first three, 3e-2 / 5e-2 for the last two, 3e-2 for the
        mean of the runs' last fifths (see the comment at the check);
      * mAP / mAP30 / mAP50: first-epoch mAP is a NOISY statistic of a chaotic trajectory.  The fixture holds 24 CPU runs of
        this very epoch: twelve of the fp32 trainer, twelve of its bf16-storage emulation - seven / five under different
        torch thread counts (summation orders), five / seven from initial weights one ulp away in ONE stem weight (an fp32 ulp
        for the fp32 trainer, a bf16 ulp for the emulation: `oracle/first_epoch.py --extra2`; the ulp draws spread twice as
        wide as the thread-count draws): mAP50 0.036 .. 0.098, mean 0.072, sigma 0.016 (fp32 0.076, emulation 0.068).  The HIP
        trainer is run EIGHT times here, under the eight summation orders its own kernels offer ({CSP main / short data
        gradients as one launch | two} x {BatchNorm-backward reduction in the data gradient | as its own pass} x {the stem's
        backward as one kernel | two launches} - EngineOptions, no other difference).  The two samples are compared by Welch's
        t (unequal variances): |t| <= 2.5 for mAP, mAP30 and mAP50 (measured about -1.7 for mAP50: 0.061 vs 0.072); every
        single run must stay above 0.4 x the CPU mean (a collapsed run) and below mean + 6 sigma; the z of the HIP mean
        against the fp32 runs and against the bf16-emulation runs is printed separately, and the latter is held to >= -2
        (one-sided: the emulation is what the HIP trainer is claimed to match, so a downward drift must fail).  (Round 4, 24 HIP trajectories on the
        final kernels - eight kernel variants + sixteen bf16-ulp draws, profiles/r04_first_epoch_samples.txt: mAP50 0.0689
        against 0.0678 for the twelve emulation runs, t = +0.2, and 0.0764 for the twelve fp32 runs, t = -1.4: the HIP trainer
        sits on the CPU trainer's bf16-storage emulation; what separates both from the fp32 trainer is bf16 storage.)
      * every variant's loss trajectory against the DEFAULT HIP run: 5e-3 rel over the first 5 steps, 3e-2 over the first
        50 - a kernel variant with a bug separates from its siblings at once, whatever the chaotic mAP says.
        Evaluating HIP-trained weights with the CPU oracle's eval pipeline reproduces the HIP mAP to 1e-4: validation
        itself is exact - DESIGN section 5.)"""
    from oracle import first_epoch as FE
    from object_detection_cib_amd.data.detection import DetectionTarget
    from object_detection_cib_amd.engine.options import EngineOptions
    g = golden("first_epoch")
    cfg = FE.CONFIG
    assert repr(sorted(cfg.items())) == str(g["config"][0]), "fixture was generated with another CONFIG: regenerate"
    S, B, nc, seed = cfg["S"], cfg["B"], cfg["nc"], cfg["seed"]
    train = synth.coco_zipf_like(cfg["n_train"], S, cfg["data_seed"], nc)
    val = synth.coco_zipf_like(cfg["n_val"], S, cfg["data_seed"] + 1, nc)
    order = FE.epoch_order(cfg)
    n_batches = len(order) // B
    vb = [(x.cuda(), tuple(DetectionTarget(torch.from_numpy(b), torch.from_numpy(l)) for b, l in tg), None)
          for x, tg in FE.validation_batches(cfg, val)]
    cpu = g["losses_fp32"][:, 3]
    keys = [str(k) for k in g["map_keys"]]
    samples = g["map_cpu_samples"]                          # [12 CPU runs, 5 metrics]
    assert samples.shape[0] >= 24
    mean, sd = samples.mean(0), samples.std(0, ddof=1)
    assert mean[keys.index("map50")] > 0.03, "the fixture epoch must leave zero for the comparison to mean anything"

    def hip_epoch(**switches):
        pipe = DeviceTrainPipeline([c[0] for c in train], [c[1] for c in train], [c[2] for c in train], S, "cuda", rng_seed=51)
        exp = _experiment(cfg["widen"], cfg["deepen"], nc, seed)
        opts = EngineOptions.from_env()
        for k, v in switches.items():
            assert hasattr(opts, k)
            setattr(opts, k, v)
        exp.net.engine_options = opts
        exp.val_nms_conf_threshold, exp.val_nms_iou_threshold = cfg["conf_thres"], cfg["nms_thres"]
        random.seed(seed); np.random.seed(seed)
        losses = []
        for step in range(n_batches):
            img, _, targets = pipe.make_batch([int(i) for i in order[step * B:(step + 1) * B]])
            losses.append(exp.optimize((img, targets, None), n_batches).detach())
        exp.end_epoch()
        hip = torch.stack(losses).cpu().numpy().astype(np.float64)
        assert np.isfinite(hip).all()
        rel = np.abs(hip - cpu) / np.abs(cpu)
        assert rel[:5].max() < 1e-2 and rel[:50].max() < 5e-2, (switches, rel[:5], rel[:50].max())
        # per-fifth means of the loss: the trajectories separate as the epoch goes on (chaotic dynamics; the eight HIP
        # summation-order variants of profiles/r03_first_epoch_samples.txt end between 2.69 and 2.85, the three CPU
        # trajectories of the fixture between 2.755 and 2.781), so a single run is held to 2 % over the first three fifths,
        # 3 % in the fourth, 5 % in the last - and the MEAN of the runs' last fifths to 3 % below
        fifth = n_batches // 5
        for k in range(5):
            a, b = hip[k * fifth:(k + 1) * fifth].mean(), cpu[k * fifth:(k + 1) * fifth].mean()
            assert abs(a - b) <= (2e-2, 2e-2, 2e-2, 3e-2, 5e-2)[k] * b, (switches, k, a, b)
        last_fifths.append(hip[4 * fifth:].mean())
        trajs.append(hip)
        rep = exp.validate(vb, nc)
        del exp, pipe
        torch.cuda.empty_cache()
        return np.array([rep[k] for k in keys])

    last_fifths, trajs = [], []
    variants = [dict(dual_dgrad=d, bn_reduce_fused=r, stem_bwd_fused=f) for d in (True, False) for r in (True, False) for f in (True, False)]
    runs = np.stack([hip_epoch(**v) for v in variants])          # variants[0] = the default configuration
    for v, tr in zip(variants[1:], trajs[1:]):
        rel = np.abs(tr - trajs[0]) / np.abs(trajs[0])
        assert rel[:5].max() <= 5e-3 and rel[:50].max() <= 3e-2, (v, rel[:5].max(), rel[:50].max())
    cpu_last = np.mean([g[k][4 * (n_batches // 5):, 3].mean() for k in ("losses_fp32", "losses_fp32_alt", "losses_bf16emu")])
    assert abs(np.mean(last_fifths) - cpu_last) <= 3e-2 * cpu_last, (last_fifths, cpu_last)
    tags = [str(t) for t in g["map_sample_tags"]]
    emu = np.array(["bf16" in t or "emu" in t for t in tags])
    assert 0 < emu.sum() < len(tags), tags
    hmean, hsd = runs.mean(0), runs.std(0, ddof=1)
    nh, ncpu = runs.shape[0], samples.shape[0]
    welch = (hmean - mean) / np.sqrt(hsd ** 2 / nh + sd ** 2 / ncpu)
    zf = lambda sub: (hmean - sub.mean(0)) / np.sqrt(hsd ** 2 / nh + sub.std(0, ddof=1) ** 2 / sub.shape[0])
    z_fp32, z_emu = zf(samples[~emu]), zf(samples[emu])
    r4 = lambda a: {k: round(float(v), 4) for k, v in zip(keys[:3], a)}
    print("first-epoch mAP  HIP runs (mAP50):", [round(float(r[keys.index("map50")]), 4) for r in runs],
          " HIP mean:", r4(hmean), " HIP sigma:", r4(hsd), " CPU mean:", r4(mean), " CPU sigma:", r4(sd),
          f" Welch t (HIP {nh} vs CPU {ncpu}):", {k: round(float(v), 2) for k, v in zip(keys[:3], welch)},
          f" vs the {int((~emu).sum())} fp32 runs:", {k: round(float(v), 2) for k, v in zip(keys[:3], z_fp32)},
          f" vs the {int(emu.sum())} bf16-emulation runs:", {k: round(float(v), 2) for k, v in zip(keys[:3], z_emu)})
    for k in ("map", "map30", "map50"):
        i = keys.index(k)
        assert abs(welch[i]) <= 2.5, (k, welch[i], hmean[i], mean[i])
        # one-sided, against the runs the HIP trainer is CLAIMED to match (the CPU trainer's bf16-storage emulation): a
        # downward drift of 15 - 20 % that the two-sided test above would let through fails here (ADVICE round 4;
        # measured about -0.9 for mAP50)
        assert z_emu[i] >= -2.0, (k, z_emu[i], hmean[i], samples[emu].mean(0)[i])
        assert (runs[:, i] >= 0.4 * mean[i]).all() and (runs[:, i] <= mean[i] + 6 * sd[i]).all(), (k, runs[:, i], mean[i], sd[i])


@pytest.mark.parametrize("scale", ["yv5n_160", "yv5s_640"])
def test_class_aware_mixup_reweighted_config_tracks_cpu_oracle(scale):
    """BASELINE configs[2] + configs[3] on one GPU, all pieces together (yv5s_640: at configs[3]'s own network and
    resolution, B = 8, eight steps; yv5n_160: 24 steps of the small form): ClassAwareSampler(dataset_info) drives the epoch
    order and (through its `sampler_indices` side channel, kod/data/detection.py:114-122) the mosaic partner choice,
    mosaic + mixup (p = 0.3) compositing on the device, BCE classification loss re-weighted with
    pos_weight = sum(count) / count_c (kod/lightning/tasks/trainer.py:54-58).  The HIP trainer's loss trajectory must
    track the CPU oracle fed the same batches under the same warm-up schedule."""
    import datetime
    from object_detection_cib_amd.data.cache import DatasetInfo, ImageMetadata, SampleInfo, TargetInfo
    from object_detection_cib_amd.data.samplers import ClassAwareSampler
    (S, steps, widen), nc, B, seed = {"yv5n_160": (160, 24, 0.25), "yv5s_640": (640, 8, 0.5)}[scale], 10, 8, 9
    cache = synth.coco_zipf_like(96, S, seed, nc)
    names = [f"c{i}" for i in range(nc)]
    meta = ImageMetadata(S, S, 3, "image/jpeg", 1)
    ds = DatasetInfo("synthetic", datetime.datetime(2023, 1, 1), names,
                     [SampleInfo(str(i), f"/nowhere/{i}.jpg", meta, [TargetInfo(tuple(b), names[int(l)]) for b, l in zip(bb, lb)])
                      for i, (_, bb, lb) in enumerate(cache)])
    present = [c for c, n in ds.get_instance_count().items() if n > 0]
    ds = ds.filter("present", present) if len(present) < nc else ds            # ClassAware needs every class to occur
    counts = np.array(list(ds.get_instance_count().values()), dtype=np.float32)
    weights = (np.sum(counts) / counts)                                        # trainer.py:54-58
    full = np.ones(nc, dtype=np.float32)
    full[[names.index(c) for c in ds.classes]] = weights
    torch.manual_seed(seed)
    sampler = ClassAwareSampler(ds)
    order = list(iter(sampler))
    pipe = DeviceTrainPipeline([c[0] for c in cache], [c[1] for c in cache], [c[2] for c in cache], S, "cuda",
                               mixup_prob=0.3, sampler_indices=sampler.sampler_indices)
    random.seed(seed); np.random.seed(seed)
    torch.manual_seed(seed)
    net = Yolov5Network(3, nc, widen_factor=widen, deepen_factor=0.33).cuda().train()
    infos = (voc_anchor_info(8), voc_anchor_info(16), voc_anchor_info(32))
    loss = Yolov5Loss(Yolov5LabelAssigner(AssignmentAnchorInfo(*infos), 4.0), Yolov5LossParams.get_default(),
                      IoUCalculator("ciou", 1e-7), full.tolist())
    exp = DefaultYolov5Experiment(net, loss, LayerwiseAnchorInfo(*infos),
                                  optimizer_warmup_updater=OptimizerWarmupUpdater(3, 0.1, 0.8, 0.937))
    torch.manual_seed(seed)
    ref = OracleYolov5(3, nc, widen, 0.33).train()
    bias, decay, norm = O.param_groups(ref)
    opt = torch.optim.SGD([dict(params=bias, weight_decay=0.0), dict(params=decay, weight_decay=5e-4),
                           dict(params=norm, weight_decay=0.0)], lr=0.01, momentum=0.937, nesterov=True)
    pw = torch.from_numpy(full)
    n_batches = len(order) // B
    nw = max(round(n_batches * 3), 100)
    hip, cpu, mixed = [], [], 0
    for step in range(steps):
        idx = [order[(step * B + k) % len(order)] for k in range(B)]
        img, _, targets = pipe.make_batch(idx)
        hip.append(exp.optimize((img, targets, None), n_batches).item())
        w = O.warmup_values(step, 0, nw)
        for pg, name in zip(opt.param_groups, O.GROUP_NAMES):
            pg["lr"], pg["momentum"] = w[name]
        opt.zero_grad()
        tot = D.train_step_total(D.yolo_loss(S, S, ref(img.cpu()), [D.Target(t.boxes, t.labels) for t in targets], pos_weight=pw), B)
        tot.backward()
        opt.step()
        cpu.append(tot.item())
    hip, cpu = np.array(hip), np.array(cpu)
    assert np.isfinite(hip).all() and np.isfinite(cpu).all()
    rel = np.abs(hip - cpu) / np.abs(cpu)
    assert rel[:5].max() < 1e-2 and rel.max() < 5e-2, rel
    assert float(full.max()) > 5.0          # the Zipf tail really is re-weighted (pos_weight = sum(count) / count_c)


def test_train_step_with_device_resident_targets_equals_prebuilt_batch():
    """A Lightning training_step hands `train_step` a tuple of DetectionTargets that already live on the GPU: their
    concatenation (torch kernels on the current stream) must be ordered before the side stream's assignment kernel.
    Same losses and gradients, bit for bit, as with a prebuilt BatchedTargets - repeated, with the main stream kept
    busy in front so that a missing dependency would show."""
    from object_detection_cib_amd.core.label_assignment.yv5 import BatchedTargets
    from object_detection_cib_amd.core.types import FeatureShape
    from object_detection_cib_amd.data.detection import DetectionTarget
    S, nc, B = 160, 10, 8
    exp = _experiment(0.25, 0.33, nc, 3)
    net, loss = exp.net, exp.loss
    x, _ = synth.batch(B, S, nc, 5)
    x = x.cuda()
    tg = synth.targets(B, S, nc, 5, nmin=4, nmax=20)
    shape = FeatureShape(width=S, height=S)
    host = tuple(DetectionTarget(b, l) for b, l in tg)
    bt = BatchedTargets.from_targets(host, torch.device("cuda", 0))
    params = list(net.parameters())
    ref_total, _ = net.train_step(x, loss, shape, bt, float(B))
    ref_g = net.engine().current_grad_arena().clone()
    ref_total = ref_total.item()
    busy = torch.empty(64 << 20, device="cuda")
    for _ in range(3):
        # fresh device tensors written by kernels queued right in front of the step
        dev_t = tuple(DetectionTarget((b.cuda() * 2.0) / 2.0, l.cuda() + 0) for b, l in tg)
        busy.normal_()
        for p in params:
            p.grad = None                 # (a kept .grad would be accumulated into, like autograd does)
        total, _ = net.train_step(x, loss, shape, dev_t, float(B))
        assert total.item() == ref_total
        assert torch.equal(net.engine().current_grad_arena(), ref_g)
