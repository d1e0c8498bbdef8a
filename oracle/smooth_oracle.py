"""CPU oracle for the flow warp / temporal fusion of parsing maps.  TEST INFRASTRUCTURE ONLY.

numpy (fp32) restatement of smooth_parsing_map.py:37-75 (`warp`: grid + flow, normalise to [-1,1],
F.grid_sample(align_corners=True) of the tensor and of ones, threshold 0.9999) and :140,155-166 (temporal and
spatial weights, centre-frame override, normalised fusion).  grid_sample itself is PyTorch's (ATen
GridSampler.h: unnormalise ((v+1)/2)*(size-1), corner weights nw = (x_se-x)(y_se-y) ..., zeros outside).
Pinned against tensors computed by the reference's OWN source lines (tests/golden/make_golden_smooth.py ->
tests/golden/smooth.npz).  Only tests/ may import this module.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def _sample_setup(flo):
    """flo (B,2,H,W) -> integer corners and the four corner weights (zero where the corner is outside)."""
    B, _, H, W = flo.shape
    xx = np.arange(W, dtype=F32)[None, None, :].repeat(H, 1)
    yy = np.arange(H, dtype=F32)[None, :, None].repeat(W, 2)
    vx = F32(2.0) * (xx + flo[:, 0]) / F32(max(W - 1, 1)) - F32(1.0)        # :58-59
    vy = F32(2.0) * (yy + flo[:, 1]) / F32(max(H - 1, 1)) - F32(1.0)
    ix = ((vx + F32(1.0)) / F32(2.0)) * F32(W - 1)
    iy = ((vy + F32(1.0)) / F32(2.0)) * F32(H - 1)
    x0, y0 = np.floor(ix), np.floor(iy)
    tx, ty = ix - x0, iy - y0
    ax, ay = (x0 + F32(1.0)) - ix, (y0 + F32(1.0)) - iy
    x0i, y0i = x0.astype(np.int64), y0.astype(np.int64)
    w = []
    for dy, dx, wy, wx in ((0, 0, ay, ax), (0, 1, ay, tx), (1, 0, ty, ax), (1, 1, ty, tx)):
        inside = (x0i + dx >= 0) & (x0i + dx < W) & (y0i + dy >= 0) & (y0i + dy < H)
        w.append((np.where(inside, wx * wy, F32(0)).astype(F32), np.clip(y0i + dy, 0, H - 1), np.clip(x0i + dx, 0, W - 1)))
    return w


def _gather(x, w):
    """x (B,C,H,W) sampled with the corner set of _sample_setup -> (B,C,H,W)."""
    B, C = x.shape[:2]
    out = np.zeros_like(x, dtype=F32)
    bi = np.arange(B)[:, None, None]
    for wt, yi, xi in w:
        for c in range(C):
            out[:, c] += x[bi, c, yi, xi] * wt
    return out


def warp(x, flo):
    """(output * mask, mask) as smooth_parsing_map.py:37-75; mask (B,1,H,W)."""
    w = _sample_setup(flo.astype(F32))
    msum = ((w[0][0] + w[1][0]) + w[2][0]) + w[3][0]
    mask = np.where(msum < F32(0.9999), F32(0), F32(1))[:, None]              # :68-69
    return _gather(x.astype(F32), w) * mask, mask


def temporal_weights(window):
    k = np.arange(2 * window + 1, dtype=F32)
    return np.exp(-(k - F32(window)) ** 2 / F32(2 * ((window + 0.5) ** 2))).astype(F32)      # :140


def fuse_window(image1, image2, parsing, flow_up, wt, center, sigma=0.2):
    """:155-165 for one centre frame: image1 (3,H,W), image2 (wn,3,H,W), parsing (wn,CP,H,W), flow_up (wn,2,H,W),
    wt (wn,) -> fused (CP,H,W)."""
    out, mask = warp(np.concatenate([image2, parsing], 1), flow_up)
    aI, aP = out[:, :3], out[:, 3:].copy()
    ws = np.exp(-((aI - image1[None]) ** 2).mean(axis=1, keepdims=True) / F32(2 * sigma ** 2)).astype(F32) * mask
    aP[center] = parsing[center]
    ws[center] = 1.0
    weights = ws * wt.reshape(-1, 1, 1, 1)
    weights = weights / weights.sum(axis=0, keepdims=True)
    return (aP * weights).sum(axis=0).astype(F32)


def downsample(x, k=(1, 3, 3, 1)):
    """Downsample(kernel, factor=2): upfirdn2d(x, k2/sum, down=2, pad=(1,1)) (model/stylegan/model.py:53-71)."""
    from oracle import vtoonify_oracle as O
    k1 = np.asarray(k, dtype=F32)
    k2 = np.outer(k1, k1)
    return O.upfirdn2d(x.astype(F32), (k2 / k2.sum()).astype(F32), up=1, down=2, pad=(1, 1))
