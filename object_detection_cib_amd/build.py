"""Builds libkodhip.so (hand-written HIP for gfx950) in-tree with hipcc.

`python -m object_detection_cib_amd.build` or `__graft_entry__.build()`.  hipcc cross-compiles
without a GPU; the resulting .so is git-ignored but travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import glob
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libkodhip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fno-fast-math", "-ffp-contract=off"]


def _digest(paths):
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode() + b"\0" + f.read())      # (not the path: the tree moves between machines)
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def source_digest() -> str:
    """Digest of the kernel sources + flags the library is built from (stamped next to the .so; the PMC evidence under
    profiles/ records the digest it was collected on)."""
    return _digest(sorted(glob.glob(os.path.join(CSRC, "*.hip"))) + sorted(glob.glob(os.path.join(CSRC, "*.h"))))


def build(force: bool = False, verbose: bool = True) -> str:
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h")))
    stamp = os.path.join(CSRC, ".build_stamp")
    dig = _digest(srcs + hdrs)
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    objs = []
    procs = []
    for s in srcs:
        o = s[:-4] + ".o"
        objs.append(o)
        procs.append((s, subprocess.Popen([HIPCC, *FLAGS, "-c", s, "-o", o], stdout=subprocess.PIPE,
                                          stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out.decode()}")
        if verbose and out.strip():
            print(out.decode(), file=sys.stderr)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    with open(stamp, "w") as f:
        f.write(dig)
    if verbose:
        print(f"built {LIB}", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
