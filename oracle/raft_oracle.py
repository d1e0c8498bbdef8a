"""CPU oracle for RAFT's correlation lookup.  TEST INFRASTRUCTURE ONLY.

numpy restatement of model/raft/alt_cuda_corr/correlation_kernel.cu:19-120 (what alt_cuda_corr.forward
computes) and of AlternateCorrBlock (model/raft/core/corr.py:63-91).  Pinned against the reference's
own pure-PyTorch CorrBlock (corr.py:12-60: all-pairs volume + F.grid_sample), which the CUDA kernel is
the memory-efficient form of (tests/golden/make_golden_raft.py -> tests/golden/raft_corr.npz).
Only tests/ may import this module.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def corr_lookup(fmap1, fmap2, coords, r):
    """fmap1 (B,H1,W1,C), fmap2 (B,H2,W2,C), coords (B,1,H1,W1,2)=(x,y) -> (B,1,(2r+1)^2,H1,W1)."""
    B, H1, W1, C = fmap1.shape
    H2, W2 = fmap2.shape[1:3]
    rd = 2 * r + 1
    out = np.zeros((B, 1, rd * rd, H1, W1), dtype=F32)
    f2p = np.zeros((B, H2 + 2 * (rd + 1), W2 + 2 * (rd + 1), C), dtype=F32)   # zero border: within_bounds, :13-16
    for b in range(B):
        for h in range(H1):
            for w in range(W1):
                x, y = coords[b, 0, h, w]
                fx, fy = int(np.floor(x)), int(np.floor(y))
                dx, dy = F32(x - np.floor(x)), F32(y - np.floor(y))
                s = np.zeros((rd + 1, rd + 1), dtype=F32)
                for iy in range(rd + 1):
                    for ix in range(rd + 1):
                        h2, w2 = fy - r + iy, fx - r + ix
                        if 0 <= h2 < H2 and 0 <= w2 < W2:
                            s[iy, ix] = np.dot(fmap1[b, h, w].astype(np.float64), fmap2[b, h2, w2].astype(np.float64))
                blend = ((1 - dy) * (1 - dx) * s[:-1, :-1] + (1 - dy) * dx * s[:-1, 1:] +
                         dy * (1 - dx) * s[1:, :-1] + dy * dx * s[1:, 1:])          # [a (y), b (x)]
                out[b, 0, :, h, w] = blend.T.reshape(-1)                            # channel = a + rd * b, :92-95
    del f2p
    return out


def avg_pool2(x):
    """F.avg_pool2d(x, 2, stride=2) on (B,H,W,C)."""
    B, H, W, C = x.shape
    x = x[:, :H // 2 * 2, :W // 2 * 2]
    return x.reshape(B, H // 2, 2, W // 2, 2, C).mean(axis=(2, 4), dtype=np.float64).astype(F32)


def alternate_corr_block(fmap1, fmap2, coords, num_levels=4, radius=4):
    """AlternateCorrBlock (corr.py:63-91).  fmap* (B,C,H,W), coords (B,2,H,W) -> (B, levels*(2r+1)^2, H, W)."""
    B, C, H, W = fmap1.shape
    f1 = np.ascontiguousarray(fmap1.transpose(0, 2, 3, 1))
    f2 = np.ascontiguousarray(fmap2.transpose(0, 2, 3, 1))
    c = np.ascontiguousarray(coords.transpose(0, 2, 3, 1)).reshape(B, 1, H, W, 2)
    outs = []
    for i in range(num_levels):
        outs.append(corr_lookup(f1, f2, (c / F32(2 ** i)).astype(F32), radius)[:, 0])
        f2 = avg_pool2(f2)
    return (np.stack(outs, axis=1).reshape(B, -1, H, W) / np.sqrt(F32(C))).astype(F32)
