"""Thin torch <-> libkodhip glue for the GPU parity tests (calls go through the C ABI)."""
from __future__ import annotations

import torch

from object_detection_cib_amd import _lib


def stream():
    return torch.cuda.current_stream().cuda_stream


def pad(n, a):
    return (n + a - 1) // a * a


def nhwc(x: torch.Tensor) -> torch.Tensor:
    """NCHW fp32 -> contiguous [B,H,W,C] bf16 on cuda."""
    return x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).cuda()


def nchw(x: torch.Tensor) -> torch.Tensor:
    return x.float().permute(0, 3, 1, 2).contiguous().cpu()


def bf(x: torch.Tensor) -> torch.Tensor:
    """Round to bf16 and back (what the HIP path stores)."""
    return x.to(torch.bfloat16).float()


def pack(weights, stem=False, ntot=None, s2=False):
    """weights: list of fp32 [N,Cin,KH,KW] tensors forming one layer (heads: 3).  Returns dict with bf16
    packs + geometry, produced by the library's pack kernel."""
    lib = _lib.lib()
    N = sum(w.shape[0] for w in weights)
    Cin, KH, KW = weights[0].shape[1:]
    # packed K axes: tap-major, every tap padded to a multiple of 32 channels (stem: 6 kernel rows x (4 pixel pairs x 8))
    Kp = 192 if stem else KH * KW * pad(Cin, 32)
    Ntot = ntot or N
    Kdp = KH * KW * pad(Ntot, 32)
    master = torch.cat([w.reshape(-1) for w in weights]).float().cuda()
    fpack = torch.zeros(Ntot * Kp, dtype=torch.bfloat16, device="cuda")
    # s2: True = the four parity-class packs, "fold" = the folded pack of kodhip_conv_dgrad_s2f
    dsize = Cin * (16 if s2 == "fold" else 9) * pad(N, 32) if s2 else Cin * Kdp
    dpack = torch.zeros(max(dsize, 8), dtype=torch.bfloat16, device="cuda")
    descs, w_off, n_off, blk = [], 0, 0, 0
    for w in weights:
        n = w.shape[0]
        descs.append([w_off, n_off * Kp, -1 if stem else 0, n, Cin, KH, KW, Kp, Kdp, Ntot, n_off, 1 if stem else ((3 if s2 == "fold" else 2) if s2 else 0), blk])
        blk += (w.numel() + 255) // 256
        w_off += w.numel()
        n_off += n
    d = torch.tensor(descs, dtype=torch.int64, device="cuda")
    _lib.check(lib.kodhip_pack_weights(master.data_ptr(), fpack.data_ptr(), dpack.data_ptr(), d.data_ptr(),
                                       len(descs), blk, stream()), "pack")
    return dict(f=fpack, d=dpack, Kp=Kp, Kdp=Kdp, N=N, Ntot=Ntot, Cin=Cin, KH=KH, KW=KW)


def conv_fwd_raw(xb, ldx_view, pk, stride, padding, out=None, ycoff=0):
    """xb: [B,H,W,ld] bf16 cuda; ldx_view=(coff, Cin).  Returns (y_raw [B,Ho,Wo,N or ld_out] bf16, stats)."""
    lib = _lib.lib()
    B, H, W, ld = xb.shape
    coff, Cin = ldx_view
    KH, KW, N = pk["KH"], pk["KW"], pk["N"]
    Ho, Wo = (H + 2 * padding - KH) // stride + 1, (W + 2 * padding - KW) // stride + 1
    if out is None:
        out = torch.zeros((B, Ho, Wo, N), dtype=torch.bfloat16, device="cuda")
    T = lib.kodhip_conv_stats_slots(B * Ho * Wo, N)
    stats = torch.zeros(2 * N * T, dtype=torch.float32, device="cuda")
    _lib.check(lib.kodhip_conv_fwd_raw(xb.data_ptr(), pk["f"].data_ptr(), out.data_ptr(), stats.data_ptr(),
                                       B, H, W, ld, coff, Cin, N, KH, KW, stride, stride, padding, padding,
                                       pk["Kp"], out.shape[-1], ycoff, stream()), "conv_fwd_raw")
    return out, stats.view(2, N, T)
