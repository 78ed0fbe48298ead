"""DescriptorProducer - the host side of the training data protocol in a worker PROCESS.

The reference feeds its trainer from DataLoader worker processes (kod/lightning/data_module.py:135-144: `num_workers`,
`persistent_workers`): the per-sample Python of DetectionDataset.__getitem__ (kod/data/detection.py:102-156) never runs
in the process that launches the training step.  Here the pixel work is a GPU kernel, so what is left for a worker is
small - index choice, the reference's RNG draw order, box arithmetic: HostProtocol (data/host_protocol.py), 8 - 9 ms
of single-core numpy per batch of 64 - but in-process it sat next to the 11 ms step and bound the loop at 0.90 of the
step rate (a producer THREAD was slower still: the GIL).  The worker owns the three generators of the protocol (`random`,
`numpy.random`, the augmentor's `default_rng(51)`) for its whole stream of batches, so the stream is bit-identical to the
in-process one, and it emits only descriptors + boxes through a shared-memory ring: no pixels, no torch, no GPU.

It is a plain child process - `python -m object_detection_cib_amd.data.producer`, a fresh interpreter that imports
numpy and this package's host-side modules only (no torch, not the trainer's `__main__`), safe before or after this
process has initialised the GPU (a fork after HIP initialisation is not) - and never touches a GPU itself.  Its job
description travels over stdin (pickle), its results through a shared-memory ring with two counters (single producer,
single consumer).  One producer per rank: every rank's stream has its own generators, as every DataLoader worker of the
reference has.  The worker dies with the trainer (PR_SET_PDEATHSIG + a parent check before every slot it fills).
"""
from __future__ import annotations

from multiprocessing import shared_memory
from typing import List, Sequence

import os

import numpy as np

from .host_protocol import SAMPLE_DESC, PackedTargets


def _layout(B: int, cap: int):
    """byte offsets of one ring slot: header | descs | mix | boxes | labels | samples | counts"""
    off, out = 0, {}
    for name, nbytes in (("header", 64), ("descs", B * 2 * SAMPLE_DESC.itemsize), ("mix", B * 2 * 4), ("boxes", cap * 32),
                         ("labels", cap * 8), ("samples", cap * 4), ("counts", B * 4)):
        out[name] = off
        off += (nbytes + 63) // 64 * 64
    out["size"] = off
    return out


def _views(buf, base: int, lay: dict, B: int, cap: int):
    f = lambda name, dtype, count: np.frombuffer(buf, dtype=dtype, count=count, offset=base + lay[name])
    return dict(header=f("header", np.int64, 8), descs=f("descs", SAMPLE_DESC, B * 2).reshape(B, 2),
                mix=f("mix", np.float32, B * 2).reshape(B, 2), boxes=f("boxes", np.float64, cap * 4).reshape(cap, 4),
                labels=f("labels", np.int64, cap), samples=f("samples", np.int32, cap), counts=f("counts", np.int32, B))


CTRL_BYTES = 64          # ring control block: int64 produced | consumed | stop


def _ctrl(buf):
    return np.frombuffer(buf, dtype=np.int64, count=3, offset=0)


def _worker(shm_name: str, slots: int, B: int, cap: int, host_args: dict, rng_seed: int, py_seed, np_seed,
            schedule: List[List[int]], parent: int):
    import random
    import time
    from .host_protocol import HostProtocol, pack_targets
    shm = shared_memory.SharedMemory(name=shm_name)
    try:
        from multiprocessing import resource_tracker          # the segment belongs to the trainer: it unlinks it
        resource_tracker.unregister(shm._name, "shared_memory")
    except Exception:      # noqa: BLE001
        pass
    ctrl = _ctrl(shm.buf)
    gone = lambda: bool(ctrl[2]) or os.getppid() != parent         # (a trainer that was killed cannot raise `stop`)
    try:
        if py_seed is not None:
            random.seed(py_seed)
        if np_seed is not None:
            np.random.seed(np_seed)
        host = HostProtocol(rng_seed=rng_seed, **host_args)
        lay = _layout(B, cap)
        for seq, idx in enumerate(schedule):
            if gone():
                return
            descs, mix, per_sample = host.batch(idx)
            pt = pack_targets(per_sample)
            while seq - int(ctrl[1]) >= slots:                     # ring full: the consumer has not freed the slot yet
                if gone():
                    return
                time.sleep(0.0005)
            if gone():
                return
            v = _views(shm.buf, CTRL_BYTES + (seq % slots) * lay["size"], lay, B, cap)
            n = len(pt.labels)
            v["header"][0], v["header"][1], v["header"][2] = seq, n, 1 if n > cap else 0
            n = min(n, cap)
            v["descs"].view(np.uint8)[...] = descs.view(np.uint8)          # (bytes, padding included: the record the kernel reads)
            v["mix"][...] = mix
            v["boxes"][:n], v["labels"][:n], v["samples"][:n] = pt.boxes[:n], pt.labels[:n], pt.samples[:n]
            v["counts"][...] = pt.counts
            del v
            # published after the slot's contents: the ring relies on x86-64's store ordering (TSO: stores of one thread
            # become visible in program order; numpy's copies above are plain stores) and on the interpreter not reordering
            # statements; the consumer reads ctrl[0] before the slot.  A weaker memory model would need a release fence here
            ctrl[0] = seq + 1
    finally:
        del ctrl
        shm.close()


def _main():
    """child entry point: the job description (a pickled dict) arrives on stdin"""
    import pickle
    import sys
    try:
        import ctypes
        ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, 9, 0, 0, 0)      # PR_SET_PDEATHSIG, SIGKILL
    except Exception:      # noqa: BLE001
        pass
    job = pickle.load(sys.stdin.buffer)
    if os.getppid() != job["parent"]:
        # the trainer died before the death signal was armed - or this interpreter was not started by it directly (a
        # `sys.executable` that is a forking shim): say so instead of leaving "producer died (exit code 0)" behind
        print(f"descriptor producer: parent is {os.getppid()}, the job names {job['parent']}; leaving", file=sys.stderr)
        sys.exit(3)
    _worker(job["shm"], job["slots"], job["B"], job["cap"], job["host_args"], job["rng_seed"], job["py_seed"], job["np_seed"],
            job["schedule"], job["parent"])


class DescriptorProducer:
    """Runs HostProtocol.batch() for `schedule` (a list of per-batch sample-index lists) in a worker process and hands the
    results over in order.  py_seed / np_seed: what the worker seeds `random` / `numpy.random` with before its first batch
    (the in-process loop seeds the same generators in the trainer's process).  max_boxes: capacity of a slot's box arrays
    (a batch with more boxes raises here, like GraphedTrainStep's own capacity check)."""

    def __init__(self, host_args: dict, batch_size: int, schedule: Sequence[Sequence[int]], rng_seed: int = 51, py_seed=None,
                 np_seed=None, max_boxes: int = 16384, slots: int = 4):
        import pickle
        import subprocess
        import sys
        assert all(len(b) == batch_size for b in schedule)
        self.B, self.cap, self.slots = int(batch_size), int(max_boxes), int(slots)
        self.lay = _layout(self.B, self.cap)
        self.shm = shared_memory.SharedMemory(create=True, size=CTRL_BYTES + self.lay["size"] * self.slots)
        self.ctrl = _ctrl(self.shm.buf)
        self.ctrl[:] = 0
        self.n, self.seq = len(schedule), 0
        root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        self.proc = subprocess.Popen([sys.executable, "-m", "object_detection_cib_amd.data.producer"], stdin=subprocess.PIPE, env=env)
        job = dict(shm=self.shm.name, slots=self.slots, B=self.B, cap=self.cap, host_args=host_args, rng_seed=rng_seed,
                   py_seed=py_seed, np_seed=np_seed, schedule=[list(map(int, b)) for b in schedule], parent=os.getpid())
        try:
            pickle.dump(job, self.proc.stdin, protocol=pickle.HIGHEST_PROTOCOL)
            self.proc.stdin.close()
        except BrokenPipeError:
            pass

    def next(self, timeout: float = 120.0):
        """(descs [B][2], mix [B][2], PackedTargets) of the next batch of the schedule - copies, the slot is free again."""
        import time
        if self.seq >= self.n:
            raise StopIteration
        t0 = time.monotonic()
        while int(self.ctrl[0]) <= self.seq:
            if self.proc.poll() is not None and int(self.ctrl[0]) <= self.seq:
                raise RuntimeError(f"the descriptor producer died (exit code {self.proc.returncode}) before batch {self.seq}")
            if time.monotonic() - t0 >= timeout:
                raise RuntimeError(f"no batch from the descriptor producer for {timeout:.0f} s")
            time.sleep(0.0002)
        v = _views(self.shm.buf, CTRL_BYTES + (self.seq % self.slots) * self.lay["size"], self.lay, self.B, self.cap)
        seq, n, overflow = (int(x) for x in v["header"][:3])
        if seq != self.seq:
            raise RuntimeError(f"descriptor ring out of order: slot holds batch {seq}, expected {self.seq}")
        if overflow:
            raise ValueError(f"{n} target boxes in batch {seq} exceed the ring's capacity {self.cap}")
        descs = v["descs"].view(np.uint8).copy().view(SAMPLE_DESC).reshape(self.B, 2)      # (a structured copy would skip the padding)
        out = (descs, v["mix"].copy(),
               PackedTargets(v["boxes"][:n].copy(), v["labels"][:n].copy(), v["samples"][:n].copy(), v["counts"].copy()))
        del v
        self.seq += 1
        self.ctrl[1] = self.seq                                    # the slot is free again
        return out

    def close(self):
        if self.shm is None:
            return
        self.ctrl[2] = 1
        try:
            self.proc.wait(5.0)
        except Exception:          # noqa: BLE001
            self.proc.kill()
            self.proc.wait(5.0)
        self.ctrl = None
        shm, self.shm = self.shm, None
        try:
            shm.close()               # (BufferError while a caller still holds views of a slot, e.g. after next() raised)
        except BufferError:
            pass
        finally:
            shm.unlink()              # the /dev/shm segment goes away whatever happened above

    def __del__(self):
        try:
            self.close()
        except Exception:          # noqa: BLE001 - interpreter shutdown
            pass


if __name__ == "__main__":
    _main()
