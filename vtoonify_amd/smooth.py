"""Temporal smoothing of parsing maps on the MI355X kernels (csrc/flow_ops.hip) -- the reference's
flicker-reduction pre-pass, smooth_parsing_map.py.

    warp(x, flo)                      smooth_parsing_map.py:37-75 (same name, arguments and return pair)
    temporal_weights(window)          :140
    fuse_window(...)                  :155-167, one centre frame: warp + spatial x temporal weights + fusion + Downsample
    smooth_parsing_maps(...)          :125-168, the loop over a video, with the optical flow supplied by a callable

The optical flow comes from a callable `flow_fn(image1, image2) -> flow_up` for a batch of frame pairs: by default
`raft_flow_fn(vtoonify_amd.raft.RAFT(...))` (the network of model/raft/core/raft.py on the same kernels, section 4.7 of
DESIGN.md); the reference's own `raft_model(..., test_mode=True)[1]` plugs in unchanged as well (its correlation
lookup then runs on vtoonify_amd.raft_corr).  GPU fp32 tensors only.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional

import torch

from . import _lib
from . import kernels as K
from . import op


def _p(t):
    return C.c_void_p(t.data_ptr() if t is not None else 0)


def warp(x: torch.Tensor, flo: torch.Tensor):
    """warp an image/tensor (im2) back to im1 according to the optical flow: x (B,C,H,W), flo (B,2,H,W) ->
    (output * mask, mask), mask (B,C,H,W) of zeros and ones (smooth_parsing_map.py:37-75)."""
    if x.dtype != torch.float32 or flo.dtype != torch.float32:
        raise _lib.VtError("smooth.warp: fp32 tensors expected")
    K._dev_ok(x, flo)
    B, Cn, H, W = x.shape
    if tuple(flo.shape) != (B, 2, H, W):
        raise _lib.VtError("smooth.warp: flo must be (B,2,H,W)")
    out = torch.empty_like(x)
    mask = torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().vt_flow_warp(_p(out), _p(mask), _p(x), _p(flo), B, Cn, H, W, K._stream(x)), "vt_flow_warp")
    return out, mask.expand(B, Cn, H, W)


def temporal_weights(window: int, device=None) -> torch.Tensor:
    """exp(-(k - window)^2 / (2 (window + 0.5)^2)), k = 0..2*window (smooth_parsing_map.py:140), shape (2w+1,)."""
    k = torch.arange(2 * window + 1, dtype=torch.float32)
    wt = torch.exp(-(k - window) ** 2 / (2 * ((window + 0.5) ** 2)))
    return wt.to(device) if device is not None else wt


def make_downsample_kernel(k=(1, 3, 3, 1), factor: int = 2) -> torch.Tensor:
    """Downsample(kernel=[1,3,3,1], factor=2).kernel (model/stylegan/model.py:53-61, make_kernel :21-29)."""
    k1 = torch.tensor(k, dtype=torch.float32)
    k2 = k1[None, :] * k1[:, None]
    return k2 / k2.sum()


def fuse_window(image1: torch.Tensor, image2: torch.Tensor, parsing: torch.Tensor, flow_up: torch.Tensor,
                wt: torch.Tensor, center_index: Optional[int] = None, sigma: float = 0.2,
                down_kernel: Optional[torch.Tensor] = None) -> torch.Tensor:
    """smooth_parsing_map.py:155-167 for one centre frame.

    image1 (3,H,W) or (1,3,H,W): the centre frame (the reference repeats it over the window);
    image2 (2w+1,3,H,W): the window's frames; parsing (2w+1,CP,H,W): their parsing maps; flow_up (2w+1,2,H,W):
    flow from the centre frame to every window frame; wt (2w+1,) or (2w+1,1,1,1).  Returns
    down(fused_Ps): (1,CP,H/2,W/2) -- or (1,CP,H,W) when down_kernel is False."""
    for t in (image1, image2, parsing, flow_up, wt):
        if t.dtype != torch.float32:
            raise _lib.VtError("smooth.fuse_window: fp32 tensors expected")
    image1 = image1.reshape(3, *image1.shape[-2:])
    wt = wt.reshape(-1)
    K._dev_ok(image1, image2, parsing, flow_up, wt)
    wn, cp, H, W = parsing.shape
    if tuple(image2.shape) != (wn, 3, H, W) or tuple(flow_up.shape) != (wn, 2, H, W) or wt.numel() != wn or \
            tuple(image1.shape) != (3, H, W):
        raise _lib.VtError("smooth.fuse_window: window tensors disagree in shape")
    ci = wn // 2 if center_index is None else int(center_index)
    fused = torch.empty((1, cp, H, W), dtype=torch.float32, device=parsing.device)
    _lib.check(_lib.lib().vt_parsing_fuse(_p(fused), _p(image2), _p(image1), _p(parsing), _p(flow_up), _p(wt), wn, ci,
                                          cp, H, W, float(sigma), K._stream(parsing)), "vt_parsing_fuse")
    if down_kernel is False:
        return fused
    kern = make_downsample_kernel().to(parsing.device) if down_kernel is None else down_kernel
    # Downsample.forward: upfirdn2d(x, kernel, up=1, down=2, pad=(p+1)//2, p//2) with p = 4 - 2 (model.py:62-71)
    return op.upfirdn2d(fused, kern, up=1, down=2, pad=(1, 1))


def raft_flow_fn(raft_model, iters: int = 20) -> Callable[[torch.Tensor, torch.Tensor], torch.Tensor]:
    """flow_fn of smooth_parsing_maps from a RAFT module (vtoonify_amd.raft.RAFT or the reference's):
    smooth_parsing_map.py:150-151 calls `raft_model((image1+1)*255/2, (image2+1)*255/2, iters=20, test_mode=True)[1]`."""
    def fn(image1: torch.Tensor, image2: torch.Tensor) -> torch.Tensor:
        return raft_model((image1 + 1) * 255.0 / 2, (image2 + 1) * 255.0 / 2, iters=iters, test_mode=True)[1]
    return fn


def smooth_parsing_maps(Is: torch.Tensor, Ps: torch.Tensor, flow_fn: Callable[[torch.Tensor, torch.Tensor], torch.Tensor],
                        window: int, sigma: float = 0.2) -> torch.Tensor:
    """The loop of smooth_parsing_map.py:125-168 over a clip already on the GPU: Is (T,3,H,W) frames in [-1,1]
    (the reference's 2x-upsampled frames), Ps (T,CP,H,W) their parsing maps, flow_fn(image1, image2) -> flow_up
    (B,2,H,W) for B frame pairs.  Returns (T,CP,H/2,W/2)."""
    Is_ = torch.cat((Is[0:window], Is, Is[-window:]), dim=0)      # :128,135 (replicate the clip's ends)
    Ps_ = torch.cat((Ps[0:window], Ps, Ps[-window:]), dim=0)
    wt = temporal_weights(window, Is.device)
    kern = make_downsample_kernel().to(Is.device)
    out = []
    for ii in range(Is.shape[0]):
        i = ii + window
        image2 = Is_[i - window:i + window + 1].contiguous()
        image1 = Is_[i:i + 1].repeat(2 * window + 1, 1, 1, 1)
        flow_up = flow_fn(image1, image2).contiguous()
        out.append(fuse_window(Is_[i].contiguous(), image2, Ps_[i - window:i + window + 1].contiguous(), flow_up, wt,
                               window, sigma, kern))
    return torch.cat(out, dim=0)
