"""EngineOptions - every switch of the HIP engine in one object.

The switches are A/B and diagnostic knobs (none changes results beyond floating-point summation order, DESIGN.md
section 7); they are read from the environment ONCE, when the engine is built, and the resulting object is what the
engine consults and what bench.py logs into its JSON line - an A/B result is attributable from the record alone.
Kernel-selection knobs that libkodhip.so reads itself (tile shapes, stride-2 forms, ...) are recorded verbatim in
`native` so the record is complete.
"""
from __future__ import annotations

import dataclasses
import os
from dataclasses import dataclass, field
from typing import Dict

# knobs libkodhip.so reads with getenv (csrc/conv_igemm.hip, csrc/conv_wgrad.hip)
NATIVE_KNOBS = ("KODHIP_NO_FAST", "KODHIP_FORCE_BM", "KODHIP_FORCE_BN", "KODHIP_S2_SEPARATE", "KODHIP_S2_FOLD_MAXC",
                "KODHIP_S2_INTERLEAVE", "KODHIP_ROW3", "KODHIP_ROW3_MODES", "KODHIP_WGRAD_DMA", "KODHIP_WGRAD_SLOTS", "KODHIP_WGRAD_ROW3", "KODHIP_STEM_ROW", "KODHIP_STEM_BWD_BLOCKS", "KODHIP_STEM_BWD_TW", "KODHIP_STEM_BWD_STREAM",
                "KODHIP_PLAN_WIDE96", "KODHIP_WGRAD_TN192", "KODHIP_WGRAD_LINEAR", "KODHIP_WGRAD_ROW3_SLOTS", "KODHIP_DEBUG_STAMPS", "KODHIP_S2F_SKIP", "KODHIP_S2F_BN_CAP",
                "KODHIP_LIB", "KODHIP_WG_CUMASK", "KODHIP_BN_U", "KODHIP_BN_GRID", "KODHIP_BN_BLOCK", "KODHIP_BN_LDS")


def _flag(name: str, default: bool) -> bool:
    v = os.environ.get(name)
    if v is None:
        return default
    return v not in ("0", "", "false", "False")


@dataclass
class EngineOptions:
    wgrad_overlap: bool = True        # KODHIP_WGRAD_OVERLAP: weight gradients on a side stream
    wgrad_streams: int = 1            # KODHIP_WGRAD_STREAMS: side streams the weight gradients rotate over (a slab scratch each)
    wgrad_fork: str = "apply"         # KODHIP_WGRAD_FORK: "apply" (event where dY is ready, captured after the dgrad) | "legacy"
    branch_overlap: bool = True       # KODHIP_BRANCH_OVERLAP: CSP short_conv branches / P3-P4 heads on side streams
    comm_overlap: bool = True         # KODHIP_COMM_OVERLAP: gradient buckets on the weight-gradient stream, own communicator
    force_collectives: bool = False   # KODHIP_FORCE_COLLECTIVES: keep the N>1 code path on a 1-rank group
    syncbn_exchange: str = "auto"     # KODHIP_SYNCBN: "rccl" | "peer" (IPC peer buffers) | "auto" (peer when every rank is on this node)
    dual_dgrad: bool = True           # KODHIP_NO_DUAL=1 switches off
    bn_reduce_fused: bool = True      # KODHIP_NO_BNRED=1 switches off
    bn_reduce_min_k: int = 0          # KODHIP_BNRED_MINK
    dx_accum_fp32: bool = False       # KODHIP_DX_FP32: multi-consumer activation gradients accumulated in fp32
    dual_wgrad: bool = True           # KODHIP_NO_DUAL_WGRAD=1 switches off: a CSP layer's main + short weight gradients in one launch
    pair_fwd: int = 0                 # KODHIP_PAIR_FWD: a CSP layer's main_conv + short_conv forward as ONE conv launch (N = 2 * mid):
                                      # 0 off (default: measured 1.5 % slower) | 1 one apply launch for both
                                      # halves | 2 the short half's apply on the side stream
    stem_bwd_fused: bool = True       # KODHIP_STEM_BWD_FUSED: the stem's BN/SiLU backward inside its weight gradient (dY never written)
    wgrad_reduce_batched: bool = False  # KODHIP_WGRAD_REDUCE=bucket: one slab-reduction launch per gradient bucket (slower: see DESIGN)
    debug_plan: bool = False          # KODHIP_DEBUG_PLAN
    max_shape_sets: int = 4           # KODHIP_MAX_SHAPE_SETS
    bucket_mb: float = 8.0
    native: Dict[str, str] = field(default_factory=dict)

    @staticmethod
    def from_env() -> "EngineOptions":
        e = os.environ
        return EngineOptions(
            wgrad_overlap=_flag("KODHIP_WGRAD_OVERLAP", True),
            wgrad_streams=int(e.get("KODHIP_WGRAD_STREAMS", "1")),
            wgrad_fork=e.get("KODHIP_WGRAD_FORK", "apply"),
            branch_overlap=_flag("KODHIP_BRANCH_OVERLAP", True),
            comm_overlap=_flag("KODHIP_COMM_OVERLAP", True),
            force_collectives=_flag("KODHIP_FORCE_COLLECTIVES", False),
            syncbn_exchange=e.get("KODHIP_SYNCBN", "auto"),
            dual_dgrad=not _flag("KODHIP_NO_DUAL", False),
            bn_reduce_fused=not _flag("KODHIP_NO_BNRED", False),
            bn_reduce_min_k=int(e.get("KODHIP_BNRED_MINK", "0")),
            dx_accum_fp32=_flag("KODHIP_DX_FP32", False),
            dual_wgrad=not _flag("KODHIP_NO_DUAL_WGRAD", False),
            pair_fwd=int(e.get("KODHIP_PAIR_FWD", "0")),
            stem_bwd_fused=_flag("KODHIP_STEM_BWD_FUSED", True),
            wgrad_reduce_batched=e.get("KODHIP_WGRAD_REDUCE", "layer") == "bucket",
            debug_plan=_flag("KODHIP_DEBUG_PLAN", False),
            max_shape_sets=int(e.get("KODHIP_MAX_SHAPE_SETS", "4")),
            native={k: e[k] for k in NATIVE_KNOBS if k in e},
        )

    def as_dict(self) -> dict:
        return dataclasses.asdict(self)
