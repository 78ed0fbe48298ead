"""Static analysis of the backward program (pure host logic, no GPU: tests/test_host_logic.py).

The engine's backward pass (engine/executor.py) walks the forward op list in reverse.  Several decisions depend only on
WHO WRITES WHICH GRADIENT BUFFER IN WHAT ORDER, which is a property of the graph (engine/graph.py), not of the data:

* which data-gradient launch is the last writer of a conv unit's output gradient (it then also carries that unit's
  BatchNorm-backward reduction, Engine._plan_bn_fusion);
* which activation gradients have several producers - autograd sums those in fp32 (kod/nn/layers/csp.py:47-56 residual
  add, :109 concat consumers, kod/nn/necks/yolov5_pafpn.py:182-199) - and how each producer takes part in an fp32
  accumulation (EngineOptions.dx_accum_fp32, include/kodhip.h `accumulate`).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Set, Tuple

from .graph import Graph


@dataclass(frozen=True)
class Write:
    pos: int                 # position in backward order
    key: Tuple[str, object]  # ("dgrad" | "res" | "head", unit name) or ("up" | "pool", index of the op in graph.ops)
    kind: str                # dgrad | res | head | up | pool
    buf: str
    lo: int
    hi: int
    unit: Optional[str]      # conv unit whose data gradient this is (dgrad only)


def backward_writes(g: Graph, dual_shorts: Iterable[str] = ()) -> Tuple[List[Write], Dict[str, int]]:
    """Gradient-buffer writes of the backward pass in execution order, and the position of every conv unit's own
    BatchNorm backward.  dual_shorts: short_conv units whose data gradient is produced by their main_conv's launch."""
    dual_shorts = set(dual_shorts)
    writes: List[Write] = []
    upos: Dict[str, int] = {}
    pos = 0
    n_ops = len(g.ops)
    for ri, op in enumerate(reversed(g.ops)):
        idx = n_ops - 1 - ri
        if op.kind == "conv":
            u = op.unit
            upos[u.name] = pos
            pos += 1
            if u.residual is not None:
                r = u.residual
                writes.append(Write(pos, ("res", u.name), "res", r.buf.name, r.coff, r.coff + r.C, None))
                pos += 1
            if not u.stem and u.name not in dual_shorts:
                writes.append(Write(pos, ("dgrad", u.name), "dgrad", u.src.buf.name, u.src.coff, u.src.coff + u.src.C, u.name))
                pos += 1
        elif op.kind == "head":
            v = op.unit.src
            writes.append(Write(pos, ("head", op.unit.name), "head", v.buf.name, v.coff, v.coff + v.C, None))
            pos += 1
        else:
            v = op.src
            writes.append(Write(pos, (op.kind, idx), op.kind, v.buf.name, v.coff, v.coff + v.C, None))
            pos += 1
    return writes, upos


@dataclass
class F32Plan:
    """Per write key: (mode, uses_shadow).  mode as in include/kodhip.h: 0 none, 1 first producer (also store fp32),
    2 later producer (add shadow, store back), 3 last producer (add shadow), 4 last producer over exact bf16 partials.
    For "up" / "pool" writers mode 3 means: read the partial from the fp32 shadow."""
    modes: Dict[Tuple[str, object], int]
    shadow_bufs: Set[str]
    zero_first: Set[str]          # shadow buffers whose first write is partial: zero-filled before it
    unsupported: Dict[str, str]   # multi-producer buffers left in bf16 accumulation, with the reason


def plan_f32_accumulation(writes: List[Write], buf_channels: Dict[str, int]) -> F32Plan:
    by_buf: Dict[str, List[Write]] = {}
    for w in writes:
        by_buf.setdefault(w.buf, []).append(w)
    modes: Dict[Tuple[str, object], int] = {w.key: 0 for w in writes}
    shadow, zero_first, unsupported = set(), set(), {}
    overlap = lambda a, b: a.lo < b.hi and b.lo < a.hi
    for buf, ws in by_buf.items():
        later_overlap = [any(overlap(w, v) for v in ws[i + 1:]) for i, w in enumerate(ws)]
        earlier_overlap = [any(overlap(w, v) for v in ws[:i]) for i, w in enumerate(ws)]
        if not any(later_overlap):
            continue                                   # every element has one producer
        # exact bf16 partials: every producer that is overlapped by a later one is a copy, each range gets at most
        # one such copy, and the accumulating producers are conv launches (the add happens in their accumulators)
        copies = [w for w, lo_ in zip(ws, later_overlap) if lo_]
        if all(w.kind == "res" for w in copies) and all((not e) or w.kind in ("dgrad", "head") for w, e in zip(ws, earlier_overlap)) \
                and all(sum(overlap(w, c) for c in copies) <= 1 for w in ws if w not in copies) \
                and not any(overlap(a, b) for i, a in enumerate(copies) for b in copies[i + 1:]):
            for w, e in zip(ws, earlier_overlap):
                if e:
                    modes[w.key] = 4
            continue
        # fp32 shadow
        reason = None
        for i, w in enumerate(ws):
            if w.kind == "res":
                reason = "a residual pass-through writes into a multi-producer buffer that needs an fp32 shadow"
            if w.kind in ("up", "pool") and later_overlap[i]:
                reason = "an upsample / pool gradient is not the last producer of its range"
        if reason:
            unsupported[buf] = reason
            continue
        shadow.add(buf)
        first = ws[0]
        full_first = first.lo == 0 and first.hi == buf_channels[buf] and first.kind in ("dgrad", "head")
        if not full_first:
            zero_first.add(buf)
        for i, w in enumerate(ws):
            if i == 0 and full_first:
                modes[w.key] = 1
            elif w.kind in ("up", "pool"):
                modes[w.key] = 3
            else:
                modes[w.key] = 2 if (later_overlap[i] or i == 0) else 3
    return F32Plan(modes, shadow, zero_first, unsupported)


def plan_dual_dgrads(g: Graph) -> Dict[str, str]:
    """{main_conv name: short_conv name} of every CSP layer whose two entry convs can share ONE data-gradient launch
    (kodhip_conv_dgrad_dual): both pointwise, same input view, same output width (kod/nn/layers/csp.py:96-111)."""
    out = {}
    for op in g.ops:
        if op.kind != "conv":
            continue
        u, v = op.unit, op.unit.sibling
        if (v is not None and u.k == v.k == 1 and u.s == v.s == 1 and u.p == v.p == 0 and u.cout == v.cout
                and (u.src.buf.name, u.src.coff, u.src.C) == (v.src.buf.name, v.src.coff, v.src.C) and u.cout % 8 == 0):
            out[u.name] = v.name
    return out


def plan_bn_reduce_fusion(g: Graph, writes: List[Write], upos: Dict[str, int], max_segments: int = 3) -> Dict[str, List[Tuple[str, int]]]:
    """For every conv unit U: the LAST write into U's output-gradient slice before U's own BatchNorm backward.  When that
    write is a conv data gradient covering the whole slice, it can carry U's BatchNorm-backward reduction in its epilogue
    (csrc/conv_igemm.hip MODE_PLAIN_BN).  Returns {writer unit name: [(producer unit name, channel offset of the
    producer's slice inside the writer's output), ...]} (at most `max_segments` per writer: the kernel's MAX_SEG)."""
    plan: Dict[str, List[Tuple[str, int]]] = {}
    for op in g.ops:
        if op.kind != "conv":
            continue
        u = op.unit
        lo, hi = u.dst.coff, u.dst.coff + u.dst.C
        cand = [w for w in writes if w.pos < upos[u.name] and w.buf == u.dst.buf.name and w.lo < hi and w.hi > lo]
        if not cand:
            continue
        last = max(cand, key=lambda w: w.pos)
        if last.unit is not None and last.lo <= lo and last.hi >= hi:
            plan.setdefault(last.unit, []).append((u.name, lo - last.lo))
    return {k: v[:max_segments] for k, v in plan.items()}
