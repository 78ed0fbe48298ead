"""N>1 path on CPU: world_size-2 gloo run of the gradient-bucket logic and the sampler sharding."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from object_detection_cib_amd.engine.ddp import plan_buckets, launch_bucket, shard_indices


def test_plan_buckets_tiles_the_arena():
    starts = [0, 64, 1088, 5184, 5248, 9344]
    for be in (1, 100, 4096, 10 ** 9):
        plan = plan_buckets(starts, 9408, be)
        assert plan[0][2] == 9408 and plan[-1][1] == 0
        for (t0, lo0, hi0), (t1, lo1, hi1) in zip(plan, plan[1:]):
            assert lo0 == hi1 and t1 < t0
        assert all(lo in starts or lo == 0 for _, lo, _ in plan)
    assert len(plan_buckets(starts, 9408, 10 ** 9)) == 1
    assert len(plan_buckets(starts, 9408, 1)) == len(starts)


def test_shard_indices_matches_distributed_sampler():
    from torch.utils.data.distributed import DistributedSampler
    data = list(range(103))
    for world in (2, 8):
        for epoch in (0, 3):
            seen = []
            for rank in range(world):
                ref = DistributedSampler(data, num_replicas=world, rank=rank, shuffle=True, seed=2023)
                ref.set_epoch(epoch)
                got = shard_indices(len(data), rank, world, seed=2023, epoch=epoch)
                assert got == list(iter(ref))
                seen += got
            assert set(seen) == set(data)
    assert shard_indices(10, 1, 4, shuffle=False) == [1, 5, 9]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(7)
        base = torch.randn(9408)
        flat = base * (rank + 1)                       # rank r holds (r+1) * base
        starts = [0, 64, 1088, 5184, 5248, 9344]
        works = []
        for trig, lo, hi in plan_buckets(starts, 9408, 2048):      # "backward": buckets finish back to front
            works.append(launch_bucket(flat, lo, hi))
        for w in works:
            w.wait()
        expect = base * sum(r + 1 for r in range(world))
        ok = torch.allclose(flat, expect, rtol=1e-6, atol=1e-6)
        # SyncBN-style fp64 sums
        s = torch.tensor([1.0 + rank, 2.0 * (rank + 1)], dtype=torch.float64)
        dist.all_reduce(s)
        ok = ok and s.tolist() == [3.0, 6.0]
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_world2_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]
