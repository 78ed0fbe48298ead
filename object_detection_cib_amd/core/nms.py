"""non_max_suppression - drop-in for kod.core.nms.non_max_suppression (kod/core/nms.py:9-75).

Same signature and result (list of [n<=300, 6] tensors: x1, y1, x2, y2, conf, cls); the boolean-mask /
nonzero / argsort / torchvision.ops.nms chain of the reference is three HIP kernels per batch
(csrc/postproc.hip), so validation stays on the device.
"""
from __future__ import annotations

from typing import Sequence

import torch

from .. import _host, _lib


_WORKSPACE = {}


def non_max_suppression(detections: torch.Tensor, conf_thres: float = 0.25, nms_thres: float = 0.45,
                        classes=None) -> Sequence[torch.Tensor]:
    _lib.require_gpu()
    det = detections.contiguous().float()
    B, rows, P = det.shape
    nc = P - 5
    if classes is not None:
        # nms.py:52-54 drops candidates whose class is not listed, before the top-30000 cut: a zeroed class
        # probability never passes `cls * obj > conf_thres`, so masking the columns is the same filter
        keep = torch.zeros(nc, dtype=torch.bool, device=det.device)
        keep[torch.as_tensor(list(classes), dtype=torch.long, device=det.device)] = True
        det = det.clone()
        det[..., 5:] *= keep.to(det.dtype)
    max_wh, max_det, max_nms = 4096.0, 300, 30000                  # nms.py:22-26
    need = rows * nc
    key_cap = 64
    while key_cap < need:
        key_cap <<= 1
    dev = det.device
    # the sort keys are the one large workspace (B * key_cap * 8 B = 134 MB at 64 x 25200 x 10): kept across calls,
    # returning it to the caching allocator every batch makes later allocations fall through to hipMalloc (tens of ms)
    # (grow-only per device: alternating full and partial validation batches reuse the largest buffer)
    keys = _WORKSPACE.get(dev)
    if keys is None or keys.numel() < B * key_cap:
        keys = _WORKSPACE[dev] = torch.empty(B * key_cap, dtype=torch.int64, device=dev)
    ncand = torch.empty(B, dtype=torch.int32, device=dev)
    out = torch.empty((B, max_det, 6), dtype=torch.float32, device=dev)
    nout = torch.empty(B, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().kodhip_nms(det.data_ptr(), keys.data_ptr(), key_cap, ncand.data_ptr(), out.data_ptr(),
                                     nout.data_ptr(), B, rows, nc, float(conf_thres), float(nms_thres), max_det,
                                     max_nms, max_wh, torch.cuda.current_stream().cuda_stream), "nms")
    counts = _host.fetch(nout)[0].tolist()                           # one polling hand-off per batch (sizes the outputs)
    return [out[b, :n] for b, n in enumerate(counts)]
