"""RAFT's memory-efficient correlation block on the MI355X kernel -- the reference's third native
extension.

`forward(fmap1, fmap2, coords, radius)` has the signature and return convention of the pybind module
`alt_cuda_corr` (model/raft/alt_cuda_corr/correlation.cpp:24-34: a list with one tensor);
`AlternateCorrBlock` is model/raft/core/corr.py:63-91 on top of it, with the 2x2 average-pool pyramid
of fmap2 built by vt_avgpool2x2.  GPU tensors only (no CPU path); fp32.
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib
from . import kernels as K


def _p(t):
    return C.c_void_p(t.data_ptr())


def forward(fmap1: torch.Tensor, fmap2: torch.Tensor, coords: torch.Tensor, radius: int, scale: float = 1.0,
            coord_scale: float = 1.0):
    """fmap1 (B,H1,W1,C), fmap2 (B,H2,W2,C), coords (B,1,H1,W1,2) -> [corr (B,1,(2r+1)^2,H1,W1)]."""
    for t in (fmap1, fmap2, coords):
        if t.dtype != torch.float32:
            raise _lib.VtError("raft_corr.forward: fp32 tensors expected")
    K._dev_ok(fmap1, fmap2, coords)           # contiguous, on the GPU (correlation.cpp:19-21)
    B, H1, W1, Cc = fmap1.shape
    if fmap2.ndim != 4 or fmap2.shape[0] != B or fmap2.shape[3] != Cc:
        raise _lib.VtError("raft_corr.forward: fmap2 must be (B,H2,W2,C)")
    if tuple(coords.shape) != (B, 1, H1, W1, 2):
        raise _lib.VtError("raft_corr.forward: coords must be (B,1,H1,W1,2)")
    rd = 2 * radius + 1
    corr = torch.empty((B, 1, rd * rd, H1, W1), dtype=torch.float32, device=fmap1.device)
    _lib.check(_lib.lib().vt_corr_lookup(_p(corr), _p(fmap1), _p(fmap2), _p(coords), B, H1, W1, fmap2.shape[1],
                                         fmap2.shape[2], Cc, radius, float(scale), float(coord_scale), K._stream(fmap1)),
               "vt_corr_lookup")
    return [corr]


def avg_pool2x2_nhwc(x: torch.Tensor) -> torch.Tensor:
    K._dev_ok(x)
    n, h, w, c = x.shape
    out = torch.empty((n, h // 2, w // 2, c), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().vt_avgpool2x2(_p(out), _p(x), n, h, w, c, K._stream(x)), "vt_avgpool2x2")
    return out


class AlternateCorrBlock:
    """Same constructor / call contract as model.raft.core.corr.AlternateCorrBlock: fmaps are NCHW."""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        self.num_levels, self.radius = num_levels, radius
        self.dim = fmap1.shape[1]
        self.fmap1 = fmap1.permute(0, 2, 3, 1).contiguous()      # layout change only (corr.py:81)
        f2 = fmap2.permute(0, 2, 3, 1).contiguous()
        self.pyramid = [f2]
        for _ in range(num_levels - 1):
            f2 = avg_pool2x2_nhwc(f2)
            self.pyramid.append(f2)

    def __call__(self, coords):
        coords = coords.permute(0, 2, 3, 1)
        B, H, W, _ = coords.shape
        c5 = coords.reshape(B, 1, H, W, 2).contiguous()
        out = []
        s = 1.0 / math.sqrt(float(self.dim))          # corr.py:91, folded into the store
        for i in range(self.num_levels):               # coords / 2**i (corr.py:84) = the kernel's coord_scale
            corr, = forward(self.fmap1, self.pyramid[i], c5, self.radius, scale=s, coord_scale=1.0 / 2 ** i)
            out.append(corr.squeeze(1))
        return torch.stack(out, dim=1).reshape(B, -1, H, W)
