"""ctypes binding of libkodhip.so (include/kodhip.h).  Fails loudly when the library is missing:
there is no CPU / PyTorch fallback for the hot path."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KODHIP_LIB", os.path.join(_HERE, "libkodhip.so"))   # override: diagnostic builds

vp, i32, i64, f32, f64, u32 = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_double, C.c_uint32


class KodAssignLevel(C.Structure):
    _fields_ = [("idx", vp), ("label", vp), ("gt", vp), ("anc", vp), ("count", vp),
                ("anchor_w", f32 * 3), ("anchor_h", f32 * 3), ("stride", i32)]


class KodDecodeLevel(C.Structure):
    _fields_ = [("raw", vp), ("h", i32), ("w", i32), ("stride", i32), ("anchor_w", f32 * 3), ("anchor_h", f32 * 3)]


class KodBnRedSeg(C.Structure):
    _fields_ = [("ch_begin", i32), ("ch_count", i32), ("raw", vp), ("ldr", i32), ("aff", vp), ("partials", vp)]


class KodLossLevel(C.Structure):
    _fields_ = [("logits", vp), ("grad", vp), ("idx", vp), ("label", vp), ("gt", vp), ("anc", vp),
                ("count", vp), ("cellmaps", vp), ("rowprev", vp), ("rowgrad", vp), ("tobj", vp),
                ("fh", i32), ("fw", i32), ("balance", f32)]


# name -> (restype, argtypes); mirrors include/kodhip.h one to one (tests/test_abi.py checks both ways)
SIGNATURES = {
    "kodhip_last_error": (C.c_char_p, []),
    "kodhip_version": (i32, []),
    "kodhip_device_count": (i32, []),
    "kodhip_nchw_to_nhwc4": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "kodhip_pack_weights": (i32, [vp, vp, vp, vp, i32, i64, vp]),
    "kodhip_pack_desc_bytes": (i32, []),
    "kodhip_conv_stats_slots": (i32, [i64, i32]),
    "kodhip_conv_fwd_raw": (i32, [vp, vp, vp, vp] + [i32] * 16 + [vp]),
    "kodhip_conv_fwd_head": (i32, [vp, vp, vp, vp] + [i32] * 9 + [vp]),
    "kodhip_conv_dgrad": (i32, [vp, vp, vp] + [i32] * 17 + [vp, vp]),
    "kodhip_conv_dgrad_s2": (i32, [vp, vp, vp] + [i32] * 10 + [vp, vp]),
    "kodhip_conv_dgrad_bnred_slots": (i32, [i32] * 13),
    "kodhip_conv_dgrad_bnred": (i32, [vp, vp, vp] + [i32] * 17 + [vp, vp, i32, i32, vp]),
    "kodhip_conv_dgrad_s2_bnred": (i32, [vp, vp, vp] + [i32] * 10 + [vp, vp, i32, i32, vp]),
    "kodhip_conv_dgrad_s2_folded": (i32, [i32, i32]),
    "kodhip_conv_dgrad_s2f": (i32, [vp, vp, vp] + [i32] * 10 + [vp, vp]),
    "kodhip_conv_dgrad_s2f_bnred_slots": (i32, [i32] * 6),
    "kodhip_conv_dgrad_s2f_bnred": (i32, [vp, vp, vp] + [i32] * 10 + [vp, vp, i32, i32, vp]),
    "kodhip_conv_dgrad_dual": (i32, [vp, vp, vp, vp, vp] + [i32] * 11 + [vp, vp]),
    "kodhip_conv_dgrad_dual_bnred_slots": (i32, [i32] * 6),
    "kodhip_conv_dgrad_dual_bnred": (i32, [vp, vp, vp, vp, vp] + [i32] * 11 + [vp, vp, i32, i32, vp]),
    "kodhip_conv_wgrad_splits": (i32, [i64, i32, i32]),
    "kodhip_conv_wgrad_splits_geo": (i32, [i32] * 14),
    "kodhip_conv_wgrad": (i32, [vp, vp, vp, vp] + [i32] * 18 + [f32, vp]),
    "kodhip_conv_wgrad_partial": (i32, [vp, vp, vp] + [i32] * 16 + [vp]),
    "kodhip_conv_wgrad_dual_splits": (i32, [i32] * 8),
    "kodhip_conv_wgrad_dual": (i32, [vp] * 6 + [i32] * 10 + [f32, vp]),
    "kodhip_stem_bwd_fused_blocks": (i32, [i32, i32, i32, i32]),
    "kodhip_stem_bwd_fused": (i32, [vp, vp, i32, i32, vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp]),
    "kodhip_wgrad_reduce_desc_bytes": (i32, []),
    "kodhip_wgrad_reduce_blocks": (i32, [i32, i32]),
    "kodhip_wgrad_reduce_batched": (i32, [vp, vp, vp, i32, i32, vp]),
    "kodhip_bn_reduce_partials": (i32, [vp, vp, i32, i32, vp]),
    "kodhip_bn_finalize": (i32, [vp, f64, vp, vp, vp, vp, f32, f32, vp, vp, vp, vp, i32, i32, vp]),
    "kodhip_bn_finalize_partials": (i32, [vp, i32, f64, vp, vp, vp, vp, f32, f32, vp, vp, vp, vp, i32, i32, vp]),
    "kodhip_bn_finalize_partials_pair": (i32, [vp, i32, f64, i32, f32, f32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, u32, u32, vp]),
    "kodhip_bn_silu_apply_pair": (i32, [vp, i32, i32, vp, vp, vp, i32, i32, vp, vp, vp, i32, i32, i64, vp]),
    "kodhip_bn_bwd_coeffs_partials": (i32, [vp, i32, f64, vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    "kodhip_bn_bwd_coeffs_partials2": (i32, [vp, i32, f64, vp, vp, vp, vp, vp, vp, i32, i32] * 2 + [vp]),
    "kodhip_bn_finalize_partials_peer": (i32, [vp, i32, f64, vp, vp, vp, vp, f32, f32, vp, vp, vp, vp, i32, i32, vp, u32, vp]),
    "kodhip_bn_bwd_coeffs_partials_peer": (i32, [vp, i32, f64, vp, vp, vp, vp, vp, vp, i32, i32, vp, u32, vp]),
    "kodhip_bn_silu_apply": (i32, [vp, i32, vp, vp, vp, i32, i32, vp, i32, i32, i64, i32, vp]),
    "kodhip_bn_bwd_slots": (i32, [i64, i32]),
    "kodhip_bn_silu_bwd_reduce": (i32, [vp, i32, i32, vp, i32, vp, vp, vp, vp, vp, i64, i32, vp]),
    "kodhip_bn_bwd_coeffs": (i32, [vp, vp, f64, vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    "kodhip_bn_silu_bwd_apply": (i32, [vp, i32, i32, vp, i32, vp, vp, vp, vp, i32, i32, i32, i64, i32, vp]),
    "kodhip_bn_act_apply": (i32, [vp, i32, vp, vp, vp, i32, i32, vp, i32, i32, i64, i32, i32, f32, vp]),
    "kodhip_bn_act_bwd_reduce": (i32, [vp, i32, i32, vp, i32, vp, vp, vp, vp, vp, i64, i32, i32, f32, vp]),
    "kodhip_bn_act_bwd_apply": (i32, [vp, i32, i32, vp, i32, vp, vp, vp, vp, i32, i32, i32, i64, i32, i32, f32, vp]),
    "kodhip_maxpool5_fwd": (i32, [vp, i32, i32, vp, i32, i32, vp, i32, i32, i32, i32, vp]),
    "kodhip_maxpool5_bwd": (i32, [vp, i32, i32, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    "kodhip_maxpool_fwd": (i32, [vp, i32, i32, vp, i32, i32, vp, i32, i32, i32, i32, i32, vp]),
    "kodhip_maxpool_bwd": (i32, [vp, i32, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
    "kodhip_upsample2x_fwd": (i32, [vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, vp]),
    "kodhip_upsample2x_bwd": (i32, [vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
    "kodhip_head_bwd_prep": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "kodhip_sgd_nesterov": (i32, [vp, vp, vp, vp, i64, vp, vp]),
    "kodhip_fill_u32": (i32, [vp, u32, i64, vp]),
    "kodhip_pull_from_host": (i32, [vp, vp, i64, vp]),
    "kodhip_debug_stamp": (i32, [vp, vp]),
    "kodhip_comm_load": (i32, [C.c_char_p]),
    "kodhip_comm_unique_id": (i32, [vp]),
    "kodhip_comm_init": (i32, [C.POINTER(vp), vp, i32, i32]),
    "kodhip_comm_destroy": (i32, [vp]),
    "kodhip_comm_allreduce_sum": (i32, [vp, vp, i64, i32, vp]),
    "kodhip_comm_allreduce_sum_to": (i32, [vp, vp, vp, i64, i32, vp]),
    "kodhip_comm_broadcast": (i32, [vp, vp, i64, i32, vp]),
    "kodhip_peer_create": (i32, [C.POINTER(vp), i32, i32, i64]),
    "kodhip_peer_export": (i32, [vp, vp]),
    "kodhip_peer_connect": (i32, [vp, vp]),
    "kodhip_peer_connect_local": (i32, [vp, vp]),
    "kodhip_peer_view_bytes": (i32, []),
    "kodhip_peer_view": (i32, [vp, vp]),
    "kodhip_peer_step_begin": (i32, [vp, vp]),
    "kodhip_peer_allreduce_f64": (i32, [vp, vp, vp, i32, u32, vp]),
    "kodhip_peer_allreduce_f64_multi": (i32, [vp, i32, vp, vp, i32, u32, vp]),
    "kodhip_peer_timed_out": (i32, [vp, C.POINTER(i32)]),
    "kodhip_peer_status": (i32, [vp, C.POINTER(i32)]),
    "kodhip_peer_destroy": (i32, [vp]),
    "kodhip_comm_group_start": (i32, []),
    "kodhip_comm_group_end": (i32, []),
    "kodhip_compose_desc_bytes": (i32, []),
    "kodhip_compose_batch": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    "kodhip_compose_color": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, f64, vp, vp, vp, i32, vp]),
    "kodhip_val_prep_desc_bytes": (i32, []),
    "kodhip_val_prep_batch": (i32, [vp, vp, vp, vp, i32, i32, vp]),
    "kodhip_decode": (i32, [C.POINTER(KodDecodeLevel), vp, i32, i32, i32, vp]),
    "kodhip_nms": (i32, [vp, vp, i32, vp, vp, vp, i32, i32, i32, f32, f32, i32, i32, f32, vp]),
    "kodhip_map_match": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, C.POINTER(f64), i32, i32, vp]),
    "kodhip_assign_targets": (i32, [vp, vp, vp, i32, i32, i32, i32, f32, C.POINTER(KodAssignLevel), vp]),
    "kodhip_iou_fwd": (i32, [vp, vp, vp, i64, i32, f32, vp]),
    "kodhip_iou_bwd": (i32, [vp, vp, vp, vp, vp, i64, i32, f32, vp]),
    "kodhip_yolo_loss": (i32, [C.POINTER(KodLossLevel), i32, i32, i32, i32, f32, f32, f32, vp, vp, vp, i32, vp,
                               i32, vp]),
    "kodhip_yolo_loss_iou": (i32, [C.POINTER(KodLossLevel), i32, i32, i32, i32, f32, f32, f32, vp, vp, vp, i32, vp,
                                   i32, i32, f32, vp]),
}

_lib = None


def lib():
    """Load (once) and return the ctypes handle; raises if libkodhip.so has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -m object_detection_cib_amd.build` "
                "(hipcc, gfx950).  There is no CPU fallback for the HIP hot path.")
        # torch first: the process must hold ONE HIP runtime - the one PyTorch-ROCm ships and has loaded - and
        # libkodhip.so binds to it by soname.  Loaded the other way round the system runtime comes in first and torch's
        # device discovery then fails ("no ROCm-capable device") in the same process.
        import torch  # noqa: F401
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
        _lib = h
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().kodhip_last_error().decode(errors="replace")
        raise RuntimeError(f"libkodhip {what} failed (rc={rc}): {msg}")


def cpu_share() -> int:
    """CPU cores this process may actually use: the scheduler affinity capped by the cgroup CPU quota (a GPU box hands
    one GPU a 16-core quota on a 256-thread host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        if os.path.exists("/sys/fs/cgroup/cpu.max"):                       # cgroup v2: "<quota> <period>" | "max <period>"
            q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if q != "max":
                n = min(n, max(1, int(q) // int(p)))
        elif os.path.exists("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):       # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
    except (OSError, ValueError):
        pass
    return n


_threads_limited = False


def limit_host_threads():
    """torch sizes its intra-op pool from the host's core count (128 threads on the GPU box).  The small host-side
    tensor ops of the data / validation path wake all of them, they spin, the cgroup's 16-core quota is used up and the
    WHOLE process is throttled for most of a 100 ms scheduler period: measured 37 ms per validation batch (9 throttled
    periods out of 38) against 13 ms with the pool sized to the quota.  Only ever lowers the thread count."""
    global _threads_limited
    if _threads_limited or os.environ.get("KODHIP_LIMIT_HOST_THREADS", "1") == "0":
        return
    _threads_limited = True
    import torch
    n = cpu_share()
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("object_detection_cib_amd: the HIP hot path needs an MI355X (no CPU fallback)")
    lib()
