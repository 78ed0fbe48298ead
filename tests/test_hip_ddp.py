"""Data-parallel path with two ranks sharing the one GPU of the test box (gloo transport, CUDA tensors):
SyncBN + bucketed gradient all-reduce must reproduce the single-process step on the concatenated batch.
(RCCL itself needs one GPU per rank; the collective CALL SITES are identical under backend "nccl".)"""
import os
import socket

import numpy as np

import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(seed):
    from object_detection_cib_amd.core.anchors.info import voc_anchor_info
    from object_detection_cib_amd.core.bbox.iou import IoUCalculator
    from object_detection_cib_amd.core.label_assignment.yv5 import Yolov5LabelAssigner, AssignmentAnchorInfo
    from object_detection_cib_amd.lightning.experiments.yv5_baseline.loss import Yolov5Loss, Yolov5LossParams
    from object_detection_cib_amd.nn.networks.yolov5 import Yolov5Network
    torch.manual_seed(seed)
    net = Yolov5Network(3, 10, widen_factor=0.25, deepen_factor=0.33).cuda().train()
    asg = Yolov5LabelAssigner(AssignmentAnchorInfo(voc_anchor_info(8), voc_anchor_info(16), voc_anchor_info(32)), 4.0)
    return net, Yolov5Loss(asg, Yolov5LossParams.get_default(), IoUCalculator("ciou", 1e-7), None)


def _data(size):
    from oracle import synth
    x, _ = synth.batch(4, size, 10, 3)
    tg = synth.targets(2, size, 10, 3, nmin=6, nmax=12)
    return x, tg + tg          # images 2,3 carry the same boxes as 0,1 => equal per-rank loss normalisers


def _run(net, loss, x, tg, size):
    from object_detection_cib_amd.core.types import FeatureShape
    from object_detection_cib_amd.data.detection import DetectionTarget
    res = net(x.cuda())
    lr = loss(FeatureShape(width=size, height=size), res, tuple(DetectionTarget(b, l) for b, l in tg))
    total = x.shape[0] * (lr.localization + lr.classification + lr.objectness)
    total.backward()
    net.engine().wait_grads()
    return total.item()


def _worker(rank, world, port, size, out, syncbn="rccl", pair_fwd=False):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["KODHIP_SYNCBN"] = syncbn             # "rccl": collectives of the group (gloo here); "peer": IPC peer buffers
    os.environ["KODHIP_PAIR_FWD"] = "1" if pair_fwd else "0"      # a CSP layer's main_conv + short_conv forward as one launch / two
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        net, loss = _build(5)
        net.configure_distributed(None, sync_batchnorm=True, bucket_mb=0.5)
        assert (net.engine().peer is not None) == (syncbn == "peer")
        x, tg = _data(size)
        _run(net, loss, x[2 * rank:2 * rank + 2], tg[2 * rank:2 * rank + 2], size)
        g = torch.cat([p.grad.flatten() for p in net.parameters()]).cpu()
        rm = net.engine().rm_arena.cpu()
        net.engine().sgd_step((0.1, 0.01, 0.01), (0.8, 0.8, 0.8), (0.0, 5e-4, 0.0), 1.0 / world)
        torch.cuda.synchronize()
        p = torch.cat([q.detach().flatten() for q in net.parameters()]).cpu()
        if net.engine().peer is not None:
            assert not net.engine().peer.timed_out()
            net.engine().peer.close()
        if rank == 0:
            torch.save(dict(g=g, rm=rm, p=p), out)
    finally:
        dist.destroy_process_group()


def test_two_rank_ddp_syncbn_equals_single_process(tmp_path):
    """256 px: every layer's per-rank row count (2 images) is then a multiple of the 128-row conv tile, so the fp32
    per-tile statistic partials of the single-process run coincide with the ranks' and only their fp64 combination
    order differs - the forward passes agree bit for bit.  (At 128 px the deepest maps hold 32 rows per rank: the split
    fp32 sums differ in the last ulp, and a single flipped bf16 rounding in this 4-image, random-init, train-mode-BN
    network moves the gradients by 20 % - chaos, not an exchange error; seen when the 3x3 kernels' summation order
    changed.)"""
    import torch.multiprocessing as mp
    size = 256
    ctx = mp.get_context("spawn")
    res = {}
    # SyncBN statistics through the group's collectives, and through IPC-mapped peer buffers (two processes mapping
    # each other's exchange buffer on the one GPU): the two transports must agree bit for bit.  The third run is the
    # KODHIP_PAIR_FWD=1 form of the peer route (a CSP layer's main_conv + short_conv as one convolution launch and one
    # exchange kernel for both units' sums - kodhip_bn_finalize_partials_pair with a peer view): it is the one the
    # single-process comparison below uses, so that form of the exchange is covered too.
    for mode, pair in (("rccl", False), ("peer", False), ("peer", True)):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        out = str(tmp_path / f"ddp_{mode}_{int(pair)}.pt")
        procs = [ctx.Process(target=_worker, args=(r, 2, port, size, out, mode, pair)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0, mode
        res[(mode, pair)] = torch.load(out)
    for k in ("g", "rm", "p"):
        assert torch.equal(res[("peer", False)][k], res[("rccl", False)][k]), k
    # single process on all 4 images, in the same forward form as the ranks (another form groups the fp32 statistic
    # partials differently: last-ulp differences that this 4-image random-init network amplifies, see the docstring)
    from object_detection_cib_amd.engine.options import EngineOptions
    rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()
    for pair in (False, True):
        got = res[("peer", pair)]
        net, loss = _build(5)
        opts = EngineOptions.from_env()
        opts.pair_fwd = 1 if pair else 0
        net.engine_options = opts
        x, tg = _data(size)
        _run(net, loss, x, tg, size)
        assert (net.engine().ustate["backbone.stages.stage1.blocks.1.main_conv"].pair is not None) == pair
        g1 = torch.cat([p.grad.flatten() for p in net.parameters()]).cpu()
        # summed DDP gradients = sum of per-rank gradients; single-process total = 4*(...) = 2x each rank's scaling
        assert rel(got["g"], g1) < 2e-2, (pair, rel(got["g"], g1))
        assert rel(got["rm"], net.engine().rm_arena.cpu()) < 1e-3
        net.engine().sgd_step((0.1, 0.01, 0.01), (0.8, 0.8, 0.8), (0.0, 5e-4, 0.0), 0.5)
        p1 = torch.cat([q.detach().flatten() for q in net.parameters()]).cpu()
        assert rel(got["p"], p1) < 1e-4, pair


def _native_worker(port, size, out):
    """One rank, own RCCL communicator (the product transport): raw collectives + a captured training step."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["KODHIP_FORCE_COLLECTIVES"] = "1"          # keep the N>1 code path on a 1-rank group
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        torch.cuda.set_device(0)
        from object_detection_cib_amd.engine.comm import RcclComm
        comm = RcclComm(None, torch.device("cuda", 0))
        a = torch.arange(1000, dtype=torch.float32, device="cuda")
        b = torch.arange(64, dtype=torch.float64, device="cuda") / 3
        a0, b0 = a.clone(), b.clone()
        comm.all_reduce(a); comm.all_reduce(b); comm.broadcast(a, 0)
        torch.cuda.synchronize()
        assert torch.equal(a, a0) and torch.equal(b, b0)
        comm.close()

        x, tg = _data(size)
        res = {}
        for mode in ("plain", "default-eager", "default-graph", "inorder-eager", "inorder-graph", "rccl-eager", "rccl-graph"):
            net, loss = _build(5)
            eng = net.engine()
            if mode.startswith("rccl"):          # SyncBN statistics as RCCL all-reduces instead of the peer exchange (default)
                eng.opt.syncbn_exchange = "rccl"
            # default-*: what a job gets without any switch - gradient buckets overlapped with backward on the
            # weight-gradient stream through their own communicator; inorder-*: KODHIP_COMM_OVERLAP=0
            assert eng.comm_overlap, "bucket / backward overlap must be the default"
            if mode.startswith("inorder"):
                eng.comm_overlap = False
            if mode != "plain":
                net.configure_distributed(None, sync_batchnorm=True, bucket_mb=0.5, native_rccl=True)
                assert eng.comm is not None and eng.collectives
                assert (eng.comm_buckets is not None) == eng.comm_overlap
                assert (eng.peer is not None) == (not mode.startswith("rccl"))
            eng.sgd_step((0.1, 0.01, 0.01), (0.8, 0.8, 0.8), (0.0, 5e-4, 0.0), 1.0)
            from object_detection_cib_amd.core.label_assignment.yv5 import BatchedTargets
            from object_detection_cib_amd.core.types import FeatureShape
            from object_detection_cib_amd.data.detection import DetectionTarget
            xb = x.cuda()
            bt = BatchedTargets.from_targets(tuple(DetectionTarget(b_, l_) for b_, l_ in tg), torch.device("cuda", 0))
            params = list(net.parameters())

            def step():
                for p in params:
                    p.grad = None
                # Yolov5Network.train_step: the route bench.py and the captured training loop take (label assignment on
                # a side stream, one pass of the loss kernels, engine backward)
                total, _ = net.train_step(xb, loss, FeatureShape(width=size, height=size), bt, 4.0)
                eng.wait_grads()
                eng.sgd_step_device()
                return total
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                losses = [step().item() for _ in range(2)]
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            if mode.endswith("-graph"):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    last = step()
                for _ in range(3):
                    g.replay()
                    losses.append(last.item())
            else:
                losses += [step().item() for _ in range(3)]
            res[mode] = (losses, torch.cat([q.detach().flatten() for q in net.parameters()]).cpu())
            if eng.peer is not None:
                assert not eng.peer.timed_out()
            for c in (eng.comm, eng.comm_buckets, eng.peer):
                if c is not None:
                    c.close()
        torch.save(res, out)
    finally:
        dist.destroy_process_group()


def test_native_rccl_comm_and_captured_step(tmp_path):
    """The RCCL transport of the product path on the single GPU of the test box: a 1-rank communicator must be an
    identity, and a training step with SyncBN sums + gradient buckets going through it - eagerly and replayed as
    one hipGraph - must reproduce the plain single-GPU trajectory bit for bit."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "native.pt")
    p = mp.get_context("spawn").Process(target=_native_worker, args=(port, 128, out))
    p.start()
    p.join(300)
    assert p.exitcode == 0
    res = torch.load(out)
    ref_l, ref_p = res["plain"]
    # default-*: gradient buckets on the weight-gradient stream through their own communicator, eager and captured;
    # inorder-*: every collective on the main stream (KODHIP_COMM_OVERLAP=0)
    # rccl-*: SyncBN statistics as RCCL all-reduces (KODHIP_SYNCBN=rccl) instead of the peer-buffer exchange
    for mode in ("default-eager", "default-graph", "inorder-eager", "inorder-graph", "rccl-eager", "rccl-graph"):
        l, prm = res[mode]
        assert l == ref_l, (mode, l, ref_l)
        assert torch.equal(prm, ref_p), mode


def test_bench_self_launch_runs_the_collective_path():
    """`python bench.py --gpus N` with WORLD_SIZE unset (the driver's plain command) starts the ranks itself.  Rehearsed
    here with one rank and KODHIP_FORCE_COLLECTIVES=1: a fresh child process, its own RCCL communicators, SyncBN sums +
    gradient buckets (overlapped with backward, the default) inside the replayed hipGraph; ONE JSON line comes back."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["KODHIP_FORCE_COLLECTIVES"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--launch", "self", "--steps", "3",
                        "--warmup", "2", "--batch", "8", "--size", "320", "--no-cpu-baseline", "--timeout", "400"],
                       capture_output=True, text=True, timeout=500, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and len(d["per_rank_images_per_sec"]) == 1
    cfg = d["config"]
    assert cfg["launcher"] == "self" and cfg["launch"] == "hipGraph replay", cfg
    assert "RCCL" in cfg["collectives"] and "overlapped with backward" in cfg["collectives"], cfg
    assert d["engine_options"]["comm_overlap"] is True and d["engine_options"]["force_collectives"] is True
    assert abs(sum(f["share_of_step"] for f in d["families"]) - 1.0) < 0.02


def test_bench_two_ranks_on_one_gpu_control_flow():
    """`python bench.py --gpus 2` end to end on the one GPU of the test box (KODHIP_BENCH_ONE_GPU=1): the parent starts two
    fresh ranks, they rendezvous, map each other's SyncBN exchange buffer through HIP IPC, train with bucketed gradient
    all-reduces (through the gloo group here: RCCL wants one GPU per rank) and rank 0 reports ONE line for the job with
    both ranks' rates.  What an 8-GPU run adds to this is RCCL and xGMI, not control flow."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["KODHIP_BENCH_ONE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                        "--batch", "4", "--size", "256", "--no-cpu-baseline", "--timeout", "400"],
                       capture_output=True, text=True, timeout=500, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and len(d["per_rank_images_per_sec"]) == 2
    assert d["config"]["launcher"] == "self" and d["config"]["parallelism"] == "dp2+syncbn"
    assert "IPC peer buffers" in d["config"]["collectives"], d["config"]
    assert d["engine_options"]["syncbn_exchange"] == "peer"
    assert abs(d["value"] - 2 * 4 * 3 / (3 * d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]      # whole-job rate over both ranks
    assert np.isfinite(d["final_loss"])
