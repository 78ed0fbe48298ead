#!/usr/bin/env python3
"""Benchmark of the MI355X-native YOLOv5-s training step (BASELINE.json metric: images/sec, 640 px).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N ...                      (WORLD_SIZE unset: starts N fresh child processes itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = forward + target assignment + loss + backward + (gradient all-reduce) + Nesterov-SGD on a
synthetic batch that is already resident in HBM (BASELINE configs[1]: yv5s, coco-zipf-like targets,
batch 64 per GPU, 640 px, bf16 storage / fp32 accumulate).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work per image, SURVEY.md 8(d) / BASELINE.md section 2 (yv5s, nc=10, 640x640)
ALGO_BYTES_PER_IMG_BF16 = 371.5e6        # 3 * (sum conv inputs + sum conv outputs) * 2 B
ALGO_FLOP_PER_IMG = 46.785e9             # fwd + dgrad + wgrad, 2*MAC
HBM_PEAK = 8.0e12                        # B/s, MI355X_MICROARCH.md chip table
VARIANTS = {"yv5n": (0.25, 0.33), "yv5s": (0.5, 0.33), "yv5m": (0.75, 0.67)}      # (widen, deepen); yv5m = BASELINE configs[4]


def algorithmic_work(widen, deepen, nc, S):
    """SURVEY 8(d) per-image work from the network's static program: bytes = 3 * (sum conv inputs + sum conv outputs)
    * 2 B over the 57 conv units + 9 head convs, FLOP = 3 * forward 2*MAC - the stem's data gradient.  Reproduces the
    survey's figures (yv5s 371.5 MB / 46.785 GFLOP, yv5m 670.4 MB / 142.845 GFLOP, yv5n 12.147 GFLOP)."""
    from object_detection_cib_amd.engine.graph import build_graph
    g = build_graph(3, nc, widen, deepen)
    sin = sout = fl = 0
    for op in g.ops:
        if op.kind == "conv":
            u = op.unit
            hin, cin, k = (S, 3, 6) if u.stem else (S // u.src.stride, u.cin, u.k)
            ho = hin // u.s
            sin += hin * hin * cin; sout += ho * ho * u.cout; fl += 2 * ho * ho * u.cout * cin * k * k
        elif op.kind == "head":
            hh = S // op.unit.stride
            for n in (4 * g.num_anchors, g.num_anchors, nc * g.num_anchors):
                sin += hh * hh * op.unit.cin; sout += hh * hh * n; fl += 2 * hh * hh * n * op.unit.cin
    stem_dgrad = 2 * (S // 2) ** 2 * g.units[0].cout * 3 * 36
    return 3.0 * (sin + sout) * 2, 3.0 * fl - stem_dgrad
FAMILY_KERNELS = {
    "conv_fwd": "conv_igemm_kernel / conv_igemm_row3_kernel <MODE_RAW> (forward conv + BN statistics, 57 layers)",
    "dgrad": "conv_igemm[_x4|_row3]_kernel<MODE_PLAIN> (data gradient)",
    "dgrad+bn_reduce": "conv_igemm[_x4|_row3]_kernel<MODE_PLAIN_BN> (data gradient + fused BatchNorm-backward reduction)",
    "wgrad": "conv_wgrad_dma_kernel + wgrad_reduce_v4_kernel (weight gradient)",
    "bn_silu_apply": "bn_silu_apply_kernel (BatchNorm + SiLU forward, residual add)",
    "bn_silu_bwd_apply": "bn_silu_bwd_apply_kernel (BatchNorm + SiLU backward)",
    "bn_finalize": "bn_finalize_fused_kernel", "bn_bwd_coeffs": "bn_bwd_coeffs_fused_kernel",
    "bn_bwd_reduce": "bn_silu_bwd_reduce_kernel",
}


def synth_batch(B, size, nc, seed, device):
    """Seeded synthetic batch of BASELINE configs[1]: U[0,1) fp32 images [B,3,size,size] and coco-zipf-like targets -
    4..30 boxes per image (mosaic of 4 sources), class ~ Zipf(1.01) (kod/data/builder.py:110-116), log-uniform box
    sizes.  (Own generator: the timed path must not touch oracle/.)"""
    import numpy as np
    from object_detection_cib_amd.data.detection import DetectionTarget
    from object_detection_cib_amd.core.label_assignment.yv5 import BatchedTargets
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 3, size, size, generator=g)
    rng = np.random.default_rng(seed)
    k = np.arange(1, nc + 1, dtype=np.float64)
    pmf = k ** (-1.01)
    pmf /= pmf.sum()
    tg = []
    for _ in range(B):
        n = int(rng.integers(4, 31))
        c = rng.uniform(0, size, (n, 2))
        wh = np.exp(rng.uniform(np.log(8 * size / 640), np.log(400 * size / 640), (n, 2)))
        b = np.concatenate((c - wh / 2, c + wh / 2), 1).clip(0, size - 1)
        b = b[((b[:, 2] - b[:, 0]) > 2) & ((b[:, 3] - b[:, 1]) > 2)]
        if b.shape[0] == 0:
            b = np.array([[size * 0.25, size * 0.25, size * 0.75, size * 0.75]])
        lab = rng.choice(nc, size=b.shape[0], p=pmf)
        tg.append(DetectionTarget(torch.from_numpy(b.astype(np.float64)), torch.from_numpy(lab.astype(np.int64))))
    return x.to(device), BatchedTargets.from_targets(tuple(tg), device)


def build(nc, device, seed=2023, widen=0.5, deepen=0.33):
    from object_detection_cib_amd.core.anchors.info import voc_anchor_info
    from object_detection_cib_amd.core.bbox.iou import IoUCalculator
    from object_detection_cib_amd.core.label_assignment.yv5 import Yolov5LabelAssigner, AssignmentAnchorInfo
    from object_detection_cib_amd.lightning.experiments.yv5_baseline.loss import Yolov5Loss, Yolov5LossParams
    from object_detection_cib_amd.nn.networks.yolov5 import Yolov5Network
    torch.manual_seed(seed)
    net = Yolov5Network(3, nc, widen_factor=widen, deepen_factor=deepen).to(device).train()
    asg = Yolov5LabelAssigner(AssignmentAnchorInfo(voc_anchor_info(8), voc_anchor_info(16), voc_anchor_info(32)), 4.0)
    loss = Yolov5Loss(asg, Yolov5LossParams.get_default(), IoUCalculator("ciou", 1e-7), None)
    return net, loss


def cpu_baseline(seconds=12.0, seconds_1t=6.0, seconds_data=5.0):
    """The CPU oracle (pure-torch fp32 restatement of the reference trainer step) timed on the host cores:
    BASELINE configs[0] = yv5s, B=2, 640 px, fwd + assigner + loss + bwd + SGD - on every core the cgroup grants (the
    headline `value` / `cores`), on ONE thread (`one_thread`), and the reference's CPU data path (mosaic + affine warp +
    HSV + flip of one 640 px training sample, numpy, one core: `data_path`).  ~25 s of CPU work in all."""
    import random
    import numpy as np
    from oracle import datapath, detection as D, optim as O, synth
    from oracle.network import OracleYolov5
    from object_detection_cib_amd._lib import cpu_share
    torch.manual_seed(2023)
    net = OracleYolov5(3, 10, 0.5, 0.33).train()
    bias, decay, norm = O.param_groups(net)
    opt = torch.optim.SGD([dict(params=bias, weight_decay=0.0), dict(params=decay, weight_decay=5e-4),
                           dict(params=norm, weight_decay=0.0)], lr=0.01, momentum=0.937, nesterov=True)
    x, tg = synth.batch(2, 640, 10, 2023)
    tg = [D.Target(b, l) for b, l in synth.targets(2, 640, 10, 2023, nmin=4, nmax=30)]

    def step():
        opt.zero_grad(set_to_none=True)
        D.train_step_total(D.yolo_loss(640, 640, net(x), tg), 2).backward()
        opt.step()

    def timed(threads, budget):
        torch.set_num_threads(threads)
        step()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget or n == 0:
            step()
            n += 1
        return n, time.perf_counter() - t0

    share = cpu_share()                    # the cores the cgroup really grants (the GPU box gives one GPU a 16-core quota)
    n, dt = timed(share, seconds)
    out = {"value": round(2 * n / dt, 3), "unit": "images/sec", "cores": share, "kind": "port",
           "sample": f"{n} train steps of yv5s B=2 640px fp32 (oracle/ CPU restatement) in {dt:.1f}s"}
    n1, dt1 = timed(1, seconds_1t)
    out["one_thread"] = {"value": round(2 * n1 / dt1, 3), "unit": "images/sec", "cores": 1,
                         "sample": f"{n1} train steps in {dt1:.1f}s, torch.set_num_threads(1)"}
    torch.set_num_threads(share)
    # the reference's per-sample CPU data path (kod/data/detection.py:102-156 with mosaic on): numpy, single core
    pool = synth.source_samples(16, 640, seed=5)
    random.seed(2023); np.random.seed(2023)
    rng = np.random.default_rng(51)
    datapath.train_sample(pool, 0, 640, rng)
    k, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds_data or k == 0:
        datapath.train_sample(pool, k % 16, 640, rng)
        k += 1
    dtd = time.perf_counter() - t0
    out["data_path"] = {"value": round(k / dtd, 3), "unit": "images/sec", "cores": 1, "kind": "port",
                        "sample": f"{k} training samples (mosaic + affine warp + HSV + flip, 640 px; numpy restatement of the "
                                  f"reference's cv2 path) in {dtd:.1f}s on one core"}
    return out


def synth_pool(n, S, nc, seed):
    """u8 image pool + boxes of the device data path: n images with U{0..255} pixels, aspect ratios from
    {4:3, 3:4, 1:1, 3:2}, longest side S, 1..9 boxes per image, class ~ Zipf(1.01).  (Own generator: the timed path must
    not touch oracle/.)"""
    import numpy as np
    rng = np.random.default_rng(seed)
    k = np.arange(1, nc + 1, dtype=np.float64)
    pmf = k ** (-1.01)
    pmf /= pmf.sum()
    imgs, boxes, labels = [], [], []
    for _ in range(n):
        rw, rh = [(4, 3), (3, 4), (1, 1), (3, 2)][int(rng.integers(0, 4))]
        w, h = (S, int(round(S * rh / rw))) if rw >= rh else (int(round(S * rw / rh)), S)
        imgs.append(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
        m = int(rng.integers(1, 10))
        bw = np.exp(rng.uniform(np.log(S / 16), np.log(S / 2.5), (m, 2)))
        c = np.stack((rng.uniform(bw[:, 0] / 2, w - bw[:, 0] / 2), rng.uniform(bw[:, 1] / 2, h - bw[:, 1] / 2)), 1)
        boxes.append(np.concatenate((c - bw / 2, c + bw / 2), 1).astype(np.float64))
        labels.append(rng.choice(nc, size=m, p=pmf).astype(np.int64))
    return imgs, boxes, labels


def loop_leg(net, loss_fn, B, S, nc, device, steps, mixup_prob=0.0, producer=True, rank=0):
    """The training LOOP BASELINE configs[1] literally names ("mosaic on"; configs[2]: mixup_prob > 0): the reference's
    sampling / RNG protocol on the host, mosaic + affine + HSV + flip (+ mixup) compositing from a u8 pool resident in HBM,
    feeding the captured step (engine/graphed.py).  producer=True (default): the host side of the protocol runs in a worker
    process (data/producer.py - the reference's DataLoader workers) that hands descriptors over through shared memory;
    False: in this process (round 3: the loop was then bound by those 8 - 9 ms of numpy per batch at 0.90 of the step rate).
    Reported beside the step rate, never as `value`."""
    import random
    import numpy as np
    from object_detection_cib_amd.data.detection import DetectionTarget
    from object_detection_cib_amd.data.device_pipeline import DeviceTrainPipeline
    from object_detection_cib_amd.data.producer import DescriptorProducer
    from object_detection_cib_amd.engine.graphed import GraphedTrainStep
    from object_detection_cib_amd import _lib
    _lib.limit_host_threads()      # the host side of the data protocol: torch's pool sized to the cgroup's CPU share
    imgs, boxes, labels = synth_pool(256, S, nc, 7 + rank)
    pipe = DeviceTrainPipeline(imgs, boxes, labels, S, device, mixup_prob=mixup_prob)
    warm = 8                       # replays before the clock starts (3 until round 4: the first default run on a box read 0.5 % low)
    schedule = [[(i * B + k) % 256 for k in range(B)] for i in range(steps + warm + 2)]
    seed = 2023 + rank
    prod = None
    if producer:
        prod = DescriptorProducer(pipe.host_args(), B, schedule, rng_seed=51, py_seed=seed, np_seed=seed, max_boxes=16384)
    else:
        random.seed(seed); np.random.seed(seed)
    gs = None

    def produce(i):
        """batch i as bf16 pixel pairs - the layout the network's first layer reads - composited on the step's own stream
        straight into the network's input buffer (measured, tools/loop_parts.py: on a side stream beside the running step the
        VALU-heavy compositing kernel costs the step more than its own 0.3 ms, and needs a 210 MB copy into the input buffer)"""
        if prod is not None:
            descs, mix, tg = prod.next()
        else:
            descs, mix, per = pipe.host.batch(schedule[i])
            tg = tuple(DetectionTarget(torch.from_numpy(bb), torch.from_numpy(lb)) for bb, lb in per)
        _, pairs = pipe.compose_host_batch(descs, mix, out_f32=False, out_pairs=True, pairs_out=gs.input_buffer() if gs is not None else None)
        return pairs, tg
    try:
        first = produce(0)
        gs = GraphedTrainStep(net, loss_fn, B, S, S, max_targets=16384, input_pairs=True).capture(first[0], first[1])
        for i in range(warm):
            gs(*produce(i + 1))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            total, _ = gs(*produce(i + 1 + warm))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    finally:
        if prod is not None:
            prod.close()
    return {"value": round(B / dt, 1), "unit": "images/sec", "ms_per_step": round(1e3 * dt, 3), "steps": steps,
            "workload": f"DeviceTrainPipeline (mosaic + affine + colour transforms p=0.01 each + HSV + flip = the reference's default AugParams, mixup p={mixup_prob}, u8 pool of 256 images in HBM, "
                        + ("host RNG protocol in a producer process (descriptors through shared memory)" if producer else "host RNG protocol in-process")
                        + "; each batch composited as bf16 pixel pairs straight into the network's input buffer) -> hipGraph replay of the training step",
            "final_loss": float(total)}


def validation_leg(net, loss_fn, B, S, nc, device, batches=79):
    """The validation loop of the same experiment (DefaultYolov5Experiment.validate: device resize / letter-box -> eval
    forward -> decode -> NMS -> mAP matching) on a u8 pool of original-size images, random-init weights (worst-case box
    counts).  Reported beside the step rate, never as `value`."""
    from object_detection_cib_amd import _lib
    from object_detection_cib_amd.core.anchors.info import voc_anchor_info
    from object_detection_cib_amd.data.device_pipeline import DeviceValPipeline
    from object_detection_cib_amd.lightning.experiments.yv5_baseline.exp import DefaultYolov5Experiment
    from object_detection_cib_amd.lightning.experiments.yv5_baseline.type_defs import LayerwiseAnchorInfo
    _lib.limit_host_threads()
    imgs, boxes, labels = synth_pool(256, 500, nc, 3)         # longest side 500 -> resized to S
    pipe = DeviceValPipeline(imgs, boxes, labels, S, device)
    exp = DefaultYolov5Experiment(net, loss_fn, LayerwiseAnchorInfo(voc_anchor_info(8), voc_anchor_info(16), voc_anchor_info(32)))

    def feed(n):
        for i in range(n):
            img, _, t = pipe.make_batch([(i * B + k) % 256 for k in range(B)])
            yield (img, t, None)
    exp.validate(feed(6), nc)              # warm-up: allocator pools, pinned staging rings, graph capture
    # One pass = a COCO-val-sized epoch (5 000 images: 79 batches of 64), as DefaultYolov5Experiment.validate runs it: the
    # interpreter's collector paused for the epoch and run once before and once after (exp._gc_paused).  Round 3 timed 12
    # batches per call - the two collections (30 ms each in this process) were a quarter of that and moved the number by
    # +-13 % from run to run (profiles/r04_bench_repeat.txt).  Two passes, the slower one is reported.
    dts = []
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        exp.validate(feed(batches), nc)
        torch.cuda.synchronize()
        dts.append((time.perf_counter() - t0) / batches)
    dt = max(dts)
    return {"value": round(B / dt, 1), "unit": "images/sec", "ms_per_batch": round(1e3 * dt, 3), "batches": batches,
            "passes_images_per_sec": [round(B / d, 1) for d in dts],
            "workload": "DeviceValPipeline (resize + letter-box of a u8 pool) -> eval forward -> decode -> NMS -> mAP matching, "
                        "random-init weights"}


def profile_step(eng, step):
    """Per-kernel durations of every family: eager steps with HIP events around each launch on its launch stream (events
    cannot be read back from inside a replayed graph); weight gradients on the main stream for these steps so that the
    families do not overlap each other.  Returns the engine's raw profile list for family_table()."""
    eng.profile = None
    ov, eng.wgrad_overlap = eng.wgrad_overlap, False
    step(); step()
    eng.profile = []
    step()
    torch.cuda.synchronize()
    prof = eng.profile
    eng.profile, eng.wgrad_overlap = None, ov
    return prof


def variant_leg(variant, B, S, nc, device, steps=12):
    """Another network scale through the same engine (yv5m = BASELINE configs[4]'s network), one GPU, hipGraph replay of
    train_step + fused SGD on a resident synthetic batch - measured like `value`, reported beside it."""
    from object_detection_cib_amd.core.types import FeatureShape
    widen, deepen = VARIANTS[variant]
    net, loss_fn = build(nc, device, widen=widen, deepen=deepen)
    eng = net.engine()
    x, targets = synth_batch(B, S, nc, 2023, device)
    shape = FeatureShape(width=S, height=S)
    eng.sgd_step((0.1, 1e-4, 1e-4), (0.8, 0.8, 0.8), (0.0, 5e-4, 0.0), 1.0)
    params = list(net.parameters())

    def step():
        for p in params:
            p.grad = None
        total, _ = net.train_step(x, loss_fn, shape, targets, float(B))
        eng.wait_grads()
        eng.sgd_step_device()
        return total
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        last = step()
    graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        graph.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    algo_bytes, _ = algorithmic_work(widen, deepen, nc, S)
    fams = [{k: r[k] for k in ("family", "launches", "ms", "share_of_step", "GB/s", "frac")} for r in family_table(profile_step(eng, step))]
    return {"value": round(B / dt, 1), "unit": "images/sec", "ms_per_step": round(1e3 * dt, 3), "steps": steps,
            "step_roofline_frac": round(B / dt * algo_bytes / 8e12, 4), "final_loss": float(last),
            "workload": f"{variant}, B={B}, {S}px, hipGraph replay of train_step + fused SGD, resident synthetic batch",
            "families": fams}


def _beat(phase: str):
    """Progress mark of a supervised rank (KODHIP_BENCH_HEARTBEAT = a file the supervisor watches): a job none of whose
    ranks has reached a new phase for STALL_S seconds is hung - the supervisor need not wait for the whole --timeout."""
    path = os.environ.get("KODHIP_BENCH_HEARTBEAT")
    if path:
        try:
            with open(path, "a") as f:
                f.write(f"{time.time():.1f} {phase}\n")
        except OSError:
            pass


# longest silent phase of a healthy rank: a cold `import torch` + engine build, 1 - 2 minutes (KODHIP_BENCH_STALL_S: tests)
STALL_S = float(os.environ.get("KODHIP_BENCH_STALL_S", "300"))


def _last_beat(paths):
    t = 0.0
    for q in paths:
        try:
            t = max(t, os.path.getmtime(q))
        except OSError:
            pass
    return t


# What a multi-rank job falls back to when it dies or hangs (a crash inside hipGraph capture or a collective is not an
# exception a rank could catch): the same job again with eager launches, then with every collective in stream order
# through RCCL.  The JSON line records the attempt that produced it (config.attempt / config.fallback).
ATTEMPTS = (
    {},
    {"KODHIP_BENCH_NO_GRAPH": "1"},
    {"KODHIP_BENCH_NO_GRAPH": "1", "KODHIP_SYNCBN": "rccl", "KODHIP_COMM_OVERLAP": "0"},
)


try:                                   # resolved at import, before any fork: the preexec hook below only makes the call
    import ctypes as _ct
    _PRCTL = _ct.CDLL("libc.so.6", use_errno=True).prctl
except Exception:                      # noqa: BLE001
    _PRCTL = None
EXIT_DEGRADED = 75                     # the job only ran down the ATTEMPTS ladder (the JSON line says "degraded": true)


def _die_with_parent():
    """preexec of a rank this process starts: the rank gets SIGKILL when this process dies, however it dies (a launcher
    that times out kills its own children, not their children - the ranks would keep the GPUs busy under the next run).
    Between fork and exec of a multi-threaded parent nothing but the system call itself runs here."""
    if _PRCTL is not None:
        _PRCTL(1, 9, 0, 0, 0)          # PR_SET_PDEATHSIG, SIGKILL


def _stop_group(proc):
    import signal
    if proc.poll() is None:
        try:
            os.killpg(proc.pid, signal.SIGKILL)       # exactly the process group this process started
        except ProcessLookupError:
            pass
    try:
        proc.wait(10)
    except Exception:
        pass


def _launch_once(args, attempt: int, timeout: float):
    """One attempt of `python bench.py --gpus N` without a launcher: N fresh ranks, a deadline, rank 0's JSON line."""
    import socket
    import subprocess
    import threading
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    import tempfile
    n = args.gpus
    procs = []
    argv = [a for a in sys.argv[1:]]
    hb_dir = tempfile.mkdtemp(prefix="kodbench_hb_")
    beats = [os.path.join(hb_dir, f"rank{r}") for r in range(n)]
    t_start = time.time()
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KODHIP_BENCH_LAUNCHER="self",
                   KODHIP_BENCH_ATTEMPT=str(attempt), KODHIP_BENCH_HEARTBEAT=beats[r], **ATTEMPTS[attempt])
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env, preexec_fn=_die_with_parent,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, start_new_session=True))
    deadline = time.monotonic() + timeout
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while True:
        codes = [q.poll() for q in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = f"rank {bad[0][0]} exited with code {bad[0][1]}"
            break
        if all(c == 0 for c in codes):
            break
        stalled = time.time() - max(_last_beat(beats), t_start) > STALL_S
        if time.monotonic() > deadline or stalled:
            failed = (f"no rank made progress for {STALL_S:.0f} s (hang?): " if stalled else f"no result after {timeout:.0f} s (hang?): ") + ", ".join(
                f"rank {r} {'running' if c is None else 'done'}" for r, c in enumerate(codes))
            break
        time.sleep(0.2)
    import shutil
    shutil.rmtree(hb_dir, ignore_errors=True)
    if failed:
        for q in procs:
            _stop_group(q)
        return None, failed
    reader.join(10)
    line = None
    for ln in (chunks[0] if chunks else b"").decode(errors="replace").splitlines():
        if ln.startswith("{"):
            line = ln
    return line, (None if line else "rank 0 printed no JSON line")


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N FRESH child processes (one per GPU, the env contract of
    torch.distributed.run) from this parent, which never touches a GPU, wait for them with a deadline, and hand rank
    0's JSON line through.  A rank that fails - or the job hanging past --timeout - ends every child; the job is then
    tried again down the ATTEMPTS ladder, and only when the last attempt fails is the exit code non-zero: a scaling
    run never waits forever and never reports a partial job."""
    n = args.gpus
    why = None
    for k in range(len(ATTEMPTS) if n > 1 else 1):
        line, why = _launch_once(args, k, args.timeout if k == 0 else min(args.timeout, 600.0))
        if line is not None:
            print(line, flush=True)
            # a number measured down the ladder is a record of what still ran, not a successful scaling run
            return 0 if (k == 0 or args.allow_fallback) else EXIT_DEGRADED
        print(f"bench.py --gpus {n}: attempt {k} ({ATTEMPTS[k] or 'as asked'}): {why}; all ranks stopped", file=sys.stderr, flush=True)
        os.environ["KODHIP_BENCH_PREV_FAILURES"] = (os.environ.get("KODHIP_BENCH_PREV_FAILURES", "") + f"attempt {k}: {why}; ").strip()
    return 124 if why and "hang" in why else 1


def supervise_rank(args) -> int:
    """Under an external launcher (the driver's `python -m torch.distributed.run ... bench.py --gpus N`) every rank this
    launcher starts becomes a supervisor that never touches a GPU: it runs the real rank as a child process, tells the
    other supervisors how that went through a TCP store of its own (MASTER_PORT + 23 on MASTER_ADDR, rank 0 hosting), and
    when any rank of the job died or the job hangs, all supervisors stop their children and start the next attempt of
    the ATTEMPTS ladder together.  Rank 0's child inherits stdout, so the job's one JSON line passes straight through.
    Any failure to set the supervision up (port taken, store unreachable) returns None: the rank then runs in-process,
    exactly as without supervision."""
    import datetime
    import signal
    import subprocess
    import tempfile
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    try:
        from torch.distributed import TCPStore
        host, port = os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29533")) + 23
        store = TCPStore(host, port, world, rank == 0, timeout=datetime.timedelta(seconds=60), wait_for_workers=False)
        store.set(f"sup/hello/{rank}", "1")
        store.wait([f"sup/hello/{r}" for r in range(world)], datetime.timedelta(seconds=60))
    except Exception as e:      # no supervision: run as a plain rank
        print(f"[bench rank {rank}] no supervision ({type(e).__name__}: {e}); running in-process", file=sys.stderr, flush=True)
        return None
    argv = [a for a in sys.argv[1:]]
    last = 1
    for k in range(len(ATTEMPTS)):
        hb = os.path.join(tempfile.gettempdir(), f"kodbench_hb_{os.getpid()}_{k}")
        env = dict(os.environ, KODHIP_BENCH_LAUNCHER="external", KODHIP_BENCH_ATTEMPT=str(k), KODHIP_BENCH_HEARTBEAT=hb, **ATTEMPTS[k])
        child = subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env, start_new_session=True,
                                 preexec_fn=_die_with_parent)
        # a launcher that gives up sends SIGTERM to ITS children - the supervisors: take the rank (and its producer) along
        for sig in (signal.SIGTERM, signal.SIGINT):
            signal.signal(sig, lambda *_a, _c=child: (_stop_group(_c), os._exit(143)))
        t_start, t_pub = time.time(), 0.0
        deadline = time.monotonic() + (args.timeout if k == 0 else min(args.timeout, 600.0)) + 30.0
        mine, verdict, stalled = None, None, False
        while verdict is None:
            if mine is None and child.poll() is not None:
                mine = child.returncode
                try:
                    store.set(f"sup/{k}/code/{rank}", str(mine))
                except Exception:
                    return mine
            try:
                if time.time() - t_pub > 5.0:         # this rank's latest progress mark, for everybody's stall test
                    t_pub = time.time()
                    store.set(f"sup/{k}/beat/{rank}", repr(max(_last_beat([hb]), t_start)))
                codes = [store.get(f"sup/{k}/code/{r}").decode() if store.check([f"sup/{k}/code/{r}"]) else None for r in range(world)]
                seen = [float(store.get(f"sup/{k}/beat/{r}").decode()) if store.check([f"sup/{k}/beat/{r}"]) else time.time() for r in range(world)]
            except Exception:       # the store's host (rank 0's supervisor) is gone: it only leaves after a verdict
                return 0 if mine == 0 else (mine or 1)
            if time.time() - max(seen) > STALL_S + 10.0:      # (same clock: one node)
                deadline = 0.0
            if any(c not in (None, "0") for c in codes):
                verdict = "failed"
            elif all(c == "0" for c in codes):
                verdict = "ok"
            elif time.monotonic() > deadline:
                stalled = True
                try:
                    store.set(f"sup/{k}/code/{rank}", "hang")    # every supervisor sees it on its next poll
                except Exception:
                    pass
                verdict = "failed"
            else:
                time.sleep(0.2)
        try:
            os.unlink(hb)
        except OSError:
            pass
        if verdict == "ok":
            # the store lives in rank 0's supervisor: it leaves last
            try:
                store.set(f"sup/{k}/bye/{rank}", "1")
                if rank == 0:
                    store.wait([f"sup/{k}/bye/{r}" for r in range(world)], datetime.timedelta(seconds=60))
            except Exception:
                pass
            return 0 if (k == 0 or args.allow_fallback) else EXIT_DEGRADED
        _stop_group(child)
        last = mine if mine not in (None, 0) else 1
        # what ended this attempt travels into the next one's JSON line: a crash and a stall are different findings
        kind = "stalled / hung" if (stalled or any(c == "hang" for c in codes)) else "a rank crashed"
        os.environ["KODHIP_BENCH_PREV_FAILURES"] = (os.environ.get("KODHIP_BENCH_PREV_FAILURES", "") + f"attempt {k}: {kind} (rank codes {codes}); ").strip()
        if rank == 0:
            print(f"bench.py --gpus {world}: attempt {k} ({ATTEMPTS[k] or 'as asked'}) failed "
                  f"(rank codes {codes}); " + ("next attempt" if k + 1 < len(ATTEMPTS) else "giving up"), file=sys.stderr, flush=True)
        # nobody starts attempt k + 1 before every rank's child of attempt k is gone (their GPU memory, their ports)
        try:
            store.set(f"sup/{k}/stopped/{rank}", "1")
            store.wait([f"sup/{k}/stopped/{r}" for r in range(world)], datetime.timedelta(seconds=120))
        except Exception:
            return last
    return last


PMC_JSON = os.path.join(ROOT, "profiles", "r06_pmc_traffic.json")


def stamp_report(eng, graph, step, file=sys.stderr):
    """KODHIP_DEBUG_STAMPS=1: device clock stamps (kodhip_debug_stamp, 100 MHz) taken inside the step - also a replayed
    hipGraph, without a profiler attached: when each weight gradient starts relative to the main chain's stamp of the same
    unit, and when the two streams end."""
    for rep in range(3):
        graph.replay() if graph is not None else step()
        torch.cuda.synchronize()
        v = eng.stamp_buf.cpu().numpy()
        t = {n: (int(v[i]) - int(v[eng.stamp_names.index("fwd_begin")])) / 100.0 for i, n in enumerate(eng.stamp_names)}
        print("[stamps rep %d, us from fwd_begin] fwd_end %.0f bwd_begin %.0f main_end %.0f wg_end %.0f bwd_end %.0f" % (
            rep, t["fwd_end"], t["bwd_begin"], t["main_end"], t.get("wg_end", -1), t["bwd_end"]), file=file)
    rows = []
    for n in eng.stamp_names:
        if n.startswith("wg:"):
            base = n[3:].split("+")[0]
            m = t.get("m:" + base)
            rows.append("%-44s wgrad at %8.0f   main chain reached the unit at %s" % (n[3:][-44:], t[n], "%8.0f (lag %6.0f)" % (m, t[n] - m) if m is not None else "-"))
    print("\n".join(rows), file=file)


def pmc_traffic(family, B, S):
    """HBM bytes per launch of a kernel family from the committed rocprofv3 PMC passes (counters cannot be read from
    inside this process; tools/pmc_traffic.py documents the collection and the gfx950 corrections).  The passes are
    only valid for the kernels they were collected on: when the recorded digest of csrc/ differs from the tree's, the
    number is stale - the bench says so on stderr and reports traffic null rather than quote it (and
    tests/test_abi.py::test_pmc_evidence_matches_kernel_sources fails in the CPU suite until the passes are re-run)."""
    if not os.path.exists(PMC_JSON) or (B, S) != (64, 640):
        return None
    with open(PMC_JSON) as f:
        d = json.load(f)
    from object_detection_cib_amd import build as kb
    if d.get("csrc_digest") != kb.source_digest():
        print(f"bench.py: {PMC_JSON} was collected on other kernel sources (digest {d.get('csrc_digest')!r} != "
              f"{kb.source_digest()!r}): roofline.traffic = null; re-run tools/collect_evidence.sh + tools/refresh_profiles.py",
              file=sys.stderr)
        return None
    fam = d.get("families", {}).get(family)
    return round(fam["traffic_bytes_per_launch"]) if fam else None


def family_table(prof):
    """Per kernel family of one eager step: launches, summed event time, algorithmic bytes (SURVEY 8d byte model for
    the convolutions; bytes actually touched for the BatchNorm / SiLU passes), achieved GB/s and fraction of 8 TB/s.
    share_of_step = family time / sum of all families' times of that SAME eager one-stream step (shares sum to 1; the
    replayed step is shorter than that sum because its families overlap)."""
    fam = {}
    for name, e0, e1, nb, *_ in prof:
        f = fam.setdefault(name, [0, 0.0, 0.0])
        f[0] += 1
        f[1] += e0.elapsed_time(e1)
        f[2] += nb
    rows = []
    total_ms = sum(v[1] for v in fam.values()) or 1.0
    for name, (n, ms, nb) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        gbs = nb / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        rows.append({"family": name, "launches": n, "ms": round(ms, 3), "share_of_step": round(ms / total_ms, 3),
                     "algorithmic_MB": round(nb / 1e6, 1), "GB/s": round(gbs, 1), "frac": round(gbs * 1e9 / HBM_PEAK, 4)})
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU")
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--variant", default="yv5s", choices=sorted(VARIANTS), help="network scale (yv5m = BASELINE configs[4])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sync-bn", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--autograd", action="store_true",
                    help="drive the step through torch autograd (net(x) -> loss -> .backward()) instead of Yolov5Network.train_step")
    ap.add_argument("--launch", default="auto", choices=("auto", "self", "none"),
                    help="auto: with WORLD_SIZE unset and --gpus > 1 this process starts the N ranks itself; self: always")
    ap.add_argument("--timeout", type=float, default=900.0, help="self-launch: seconds before a hung job is stopped")
    ap.add_argument("--allow-fallback", action="store_true",
                    help=f"a result measured down the fallback ladder (\"degraded\": true) exits 0 instead of {EXIT_DEGRADED}")
    ap.add_argument("--no-loop", action="store_true", help="skip the training-loop leg (device data pipeline -> captured step)")
    ap.add_argument("--no-extra", action="store_true", help="skip the validation-loop and yv5m legs")
    ap.add_argument("--loop-steps", type=int, default=30)
    ap.add_argument("--loop-mixup", type=float, default=None, help="mixup probability of the loop leg (default 0 at N=1, 0.1 at N>1)")
    ap.add_argument("--loop-in-process", action="store_true", help="loop leg: host side of the data protocol in this process (round 3)")
    args = ap.parse_args()

    # decided before anything touches a GPU; the parent only starts and supervises the ranks (never re-executes itself)
    if "WORLD_SIZE" not in os.environ and (args.launch == "self" or (args.launch == "auto" and args.gpus > 1)):
        raise SystemExit(self_launch(args))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    attempt = int(os.environ.get("KODHIP_BENCH_ATTEMPT", "0"))
    # a rank started by an external launcher supervises the real rank (a child process) instead of being it - decided
    # before anything touches a GPU (KODHIP_BENCH_SUPERVISE=0: plain rank, as in rounds 1-3)
    if (world > 1 and "KODHIP_BENCH_LAUNCHER" not in os.environ and os.environ.get("KODHIP_BENCH_SUPERVISE", "1") != "0"):
        rc = supervise_rank(args)
        if rc is not None:
            raise SystemExit(rc)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP hot path has no CPU fallback)")
    _beat("start")
    # KODHIP_BENCH_ONE_GPU=1: every rank on GPU 0 (rehearsal of the multi-rank control flow on a one-GPU box: RCCL wants one
    # GPU per rank, so gradient buckets then go through the gloo group and the step runs eagerly; SyncBN still takes the
    # peer-buffer exchange, the ranks being processes of one node)
    one_gpu = os.environ.get("KODHIP_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    # KODHIP_FORCE_COLLECTIVES=1: rehearse the N>1 code path (RCCL group, SyncBN sums, gradient buckets) on one GPU
    use_dist = world > 1 or os.environ.get("KODHIP_FORCE_COLLECTIVES") == "1" or "WORLD_SIZE" in os.environ and args.launch == "self"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # control plane (rendezvous id, barriers, max-over-ranks of the clock) on gloo; every GPU collective of the
        # step goes through the engine's own RCCL communicator, enqueued on the step's streams and captured with it
        if attempt > 0 and os.environ.get("KODHIP_BENCH_LAUNCHER") == "external":
            # a later attempt under the launcher's own store: a key space of its own (attempt 0's rendezvous keys are still there)
            import datetime
            agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True"
            base = dist.TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]), world, (rank == 0) and not agent,
                                 timeout=datetime.timedelta(seconds=300), multi_tenant=True)
            dist.init_process_group("gloo", store=dist.PrefixStore(f"kodbench_attempt{attempt}", base), rank=rank, world_size=world)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")

    if use_dist:
        # a rank that stops making progress (a peer died, a collective hangs) ends itself: under a launcher without a
        # supervising parent (torch.distributed.run) the job then fails instead of hanging
        import threading
        threading.Thread(target=lambda: (time.sleep(args.timeout), print(f"[bench rank {rank}] no result after "
                         f"{args.timeout:.0f} s: giving up", file=sys.stderr, flush=True), os._exit(124)), daemon=True).start()

    nc, B, S = 10, args.batch, args.size
    widen, deepen = VARIANTS[args.variant]
    net, loss_fn = build(nc, device, widen=widen, deepen=deepen)
    algo_bytes, algo_flop = algorithmic_work(widen, deepen, nc, S)
    _beat("network built")
    if args.variant == "yv5s" and S == 640:
        assert abs(algo_bytes - ALGO_BYTES_PER_IMG_BF16) < 1e-3 * ALGO_BYTES_PER_IMG_BF16 and abs(algo_flop - ALGO_FLOP_PER_IMG) < 1e-3 * ALGO_FLOP_PER_IMG
    eng = net.engine()
    if use_dist:
        net.configure_distributed(None, sync_batchnorm=not args.no_sync_bn, native_rccl=not one_gpu)
        _beat("collectives configured")
    from object_detection_cib_amd.core.types import FeatureShape
    x, targets = synth_batch(B, S, nc, 2023 + rank, device)
    shape = FeatureShape(width=S, height=S)
    # hyper-parameters of global step 0 of the reference schedule (warm-up start, warmup.py:39-58)
    lr, mom, wd = (0.1, 0.0, 0.0), (0.8, 0.8, 0.8), (0.0, 5e-4, 0.0)
    lr = (0.1, 1e-4, 1e-4)    # non-zero so every parameter really moves

    eng.sgd_step(lr, mom, wd, 1.0 / world)        # uploads the hyper-parameters to the device buffer
    params = list(net.parameters())

    def step():
        for p in params:
            p.grad = None
        if args.autograd:          # the reference's own call sequence through torch autograd (same numbers, more launches)
            res = net(x)
            lr_ = loss_fn(shape, res, targets)
            total = B * (lr_.localization + lr_.classification + lr_.objectness)
            total.backward()
        else:
            total, _ = net.train_step(x, loss_fn, shape, targets, float(B))
        eng.wait_grads()
        eng.sgd_step_device()
        return total

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    use_graph = not args.no_graph and not (one_gpu and use_dist)      # (gloo collectives cannot be captured)
    use_graph = use_graph and os.environ.get("KODHIP_BENCH_NO_GRAPH") != "1"      # the ATTEMPTS ladder's eager rungs
    # test hook: the rank named here dies in attempt 0 (rehearsal of the ladder, tests/test_hip_ddp.py)
    if attempt == 0 and os.environ.get("KODHIP_BENCH_TEST_DIE_RANK") == str(rank) and world > 1:
        os._exit(3)
    if attempt == 0 and os.environ.get("KODHIP_BENCH_TEST_HANG_RANK") == str(rank) and world > 1:
        time.sleep(3600)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(max(args.warmup, 2) if use_graph else args.warmup):
            last = step()
    torch.cuda.current_stream().wait_stream(side)
    graph = None
    _beat("warm-up steps done")
    launch_note = "eager"
    if use_graph:
        # the whole step (~650 launches, and for N>1 the RCCL all-reduces between them) becomes one hipGraph
        barrier()
        err = None
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                last = step()
        except Exception as e:      # capture refused (e.g. a collective that cannot be captured)
            err, graph = e, None
            if not use_dist:
                raise
        # replay or eager must be ONE decision for the whole job: a rank replaying a graph while another runs eagerly
        # would issue different RCCL call sequences and hang.  Agree on it over the gloo control plane.
        ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32)
        if dist is not None:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            graph.replay()
            launch_note = "hipGraph replay"
        else:
            if err is None:
                print(f"[bench rank {rank}] another rank failed to capture the step: running eagerly everywhere",
                      file=sys.stderr, flush=True)
            else:
                print(f"[bench rank {rank}] hipGraph capture failed, running eagerly: {err}", file=sys.stderr, flush=True)
            graph = None
            torch.cuda.synchronize()
            eng._pending = []
            launch_note = "eager (graph capture failed" + (f": {type(err).__name__})" if err is not None else " on another rank)")
            for _ in range(2):
                last = step()
    barrier()
    t0 = time.perf_counter()
    _beat("timed region begins")
    for _ in range(args.steps):
        if graph is not None:
            graph.replay()
        else:
            last = step()
    barrier()
    dt = time.perf_counter() - t0
    _beat("timed region done")
    if eng.stamps_on:          # KODHIP_DEBUG_STAMPS=1: where the step's time goes, from device clock stamps
        stamp_report(eng, graph, step)
    # per-kernel durations of every family: one extra eager step with HIP events around each launch on its launch
    # stream, outside the timed region (events cannot be read back from inside a replayed graph); weight gradients
    # on the main stream for this step so that the families do not overlap each other
    prof = profile_step(eng, step)
    if eng.peer is not None and eng.peer.timed_out():
        raise SystemExit(f"[bench rank {rank}] a SyncBN peer exchange gave up waiting for another rank: the run is invalid")
    per_rank = [round(B * args.steps / dt, 1)]
    if dist is not None:
        dts = [None] * world
        dist.all_gather_object(dts, dt)
        per_rank = [round(B * args.steps / d, 1) for d in dts]
        dt = max(dts)                              # the job is as fast as its slowest rank
    final_loss = float(last.item())
    loop = None
    _beat("profile step done")
    # (with collectives in the step every rank runs the leg - its replays contain the same RCCL calls - fed by its own producer)
    if not args.no_loop and not args.autograd and graph is not None:
        mix_p = args.loop_mixup if args.loop_mixup is not None else (0.1 if world > 1 else 0.0)      # configs[2]: "mosaic+mixup"
        barrier()
        try:            # (a leg that fails is reported as its error, the measured line stands - like the extra legs below)
            loop = loop_leg(net, loss_fn, B, S, nc, device, args.loop_steps, mixup_prob=mix_p, producer=not args.loop_in_process, rank=rank)
        except Exception as e:          # noqa: BLE001
            loop = {"error": f"{type(e).__name__}: {e}"[:300]}
            torch.cuda.synchronize()
        if dist is not None:
            lts = [None] * world
            dist.all_gather_object(lts, loop.get("ms_per_step"))
            if all(t is not None for t in lts):
                loop["per_rank_ms_per_step"] = lts
                loop["ms_per_step"] = max(lts)
                loop["value"] = round(world * B / (max(lts) * 1e-3), 1)
            elif "error" not in loop:
                loop = {"error": "the loop leg failed on another rank", "per_rank_ms_per_step": lts}
    # further legs of the default single-GPU run (each reported beside `value`, none inside the timed region; a leg that
    # fails is reported as its error, the line itself stands): the validation loop and the yv5m scale
    extra = {}
    _beat("loop leg done")
    if world == 1 and not use_dist and not (args.no_extra or args.no_loop) and not args.autograd and use_graph and args.variant == "yv5s":
        for name, fn in (("validation", lambda: validation_leg(net, loss_fn, B, S, nc, device)),
                         ("yv5m", lambda: variant_leg("yv5m", B, S, nc, device))):
            try:
                extra[name] = fn()
            except Exception as e:          # noqa: BLE001 - the leg's failure must not take the measured line with it
                extra[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
            torch.cuda.synchronize()

    if rank == 0:
        ips = world * B * args.steps / dt
        # roofline: the kernel family with the largest share of the step (event-timed, see above)
        table = family_table(prof)
        top = table[0]
        n_launch = max(top["launches"], 1)
        out = {
            "metric": f"images/sec YOLOv5{args.variant[-1]} {S}px train", "value": round(ips, 2), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            # true = measured down the fallback ladder (eager launches / RCCL SyncBN) after the job as asked died or hung:
            # not evidence for the replayed-graph configuration; the launcher then exits EXIT_DEGRADED unless --allow-fallback
            "degraded": attempt > 0, "earlier_attempts": os.environ.get("KODHIP_BENCH_PREV_FAILURES") or None,
            "config": {"workload": f"{args.variant} coco-zipf-like synthetic, {S}px, bf16 storage/fp32 accumulate, "
                                   f"batch {B}/GPU, fwd+assign+loss+bwd+SGD, targets 4-30 boxes/img",
                       "global_batch": world * B, "parallelism": f"dp{world}" + ("+syncbn" if use_dist and not args.no_sync_bn else ""),
                       "launch": launch_note,
                       "collectives": ("none" if not (use_dist and eng.collectives) else
                                       ("SyncBN sums over IPC peer buffers (in the BatchNorm kernels)" if eng.peer is not None else
                                        "RCCL, native: SyncBN sums in stream order" if eng.sync_bn else "no SyncBN")
                                       + (", RCCL gradient buckets " if eng.comm is not None else ", gradient buckets through torch.distributed ")
                                       + ("overlapped with backward on the weight-gradient stream (own communicator)"
                                          if eng.comm_buckets is not None else "in stream order")),
                       "launcher": os.environ.get("KODHIP_BENCH_LAUNCHER", "external" if "WORLD_SIZE" in os.environ else "none"),
                       "attempt": attempt, "fallback": ATTEMPTS[attempt] if attempt < len(ATTEMPTS) else None},
            "per_rank_images_per_sec": per_rank,
            "engine_options": eng.opt.as_dict(),
            "final_loss": final_loss,
            "step_roofline": {"bound": "hbm", "algorithmic_bytes_per_img": algo_bytes,
                              "achieved": round(ips / world * algo_bytes / 1e9, 1), "peak": HBM_PEAK / 1e9,
                              "unit": "GB/s", "frac": round(ips / world * algo_bytes / HBM_PEAK, 4),
                              "tflops": round(ips / world * algo_flop / 1e12, 1)},
            "roofline": {"kernel": FAMILY_KERNELS.get(top["family"], top["family"]), "family": top["family"], "bound": "hbm",
                         "achieved": top["GB/s"], "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": top["frac"],
                         "traffic": pmc_traffic(top["family"], B, S) if args.variant == "yv5s" else None,
                         "avg_launch_us": round(1e3 * top["ms"] / n_launch, 2), "launches": top["launches"],
                         "algorithmic_bytes_per_launch_avg": round(1e6 * top["algorithmic_MB"] / n_launch),
                         "share_of_step": top["share_of_step"]},
            "families": table,
        }
        if loop is not None:
            out["loop"] = loop
        out.update(extra)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        for c in (eng.comm, eng.comm_buckets):
            if c is not None:
                c.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
