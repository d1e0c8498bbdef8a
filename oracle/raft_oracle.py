"""CPU oracle for RAFT: the correlation lookup and (second half of the file) the whole network.  TEST INFRASTRUCTURE ONLY.

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


# ---------------------------------------------------------------------------------------------------------
# The RAFT network (model/raft/core/raft.py:86-144 forward; extractor.py:6-60,115-183; update.py:6-139).
# numpy restatement, pinned against tensors of the reference's own RAFT class (tests/golden/raft_net.npz, made by
# tests/golden/make_golden_raft_net.py).  The correlation is the memory-efficient form above (equal to CorrBlock up
# to summation order, which is what the reference's default configuration runs).
# ---------------------------------------------------------------------------------------------------------
def _conv(x, sd, key, stride=1, pad=(0, 0)):
    from oracle import vtoonify_oracle as O
    xp = np.pad(x.astype(F32), ((0, 0), (0, 0), (pad[0], pad[0]), (pad[1], pad[1])))
    return O.conv2d(xp, sd[key + ".weight"], sd[key + ".bias"], stride, 0, 1)


def _norm(x, sd, key, kind):
    if kind == "instance":     # nn.InstanceNorm2d: affine-free, eps 1e-5, biased variance
        m = x.mean(axis=(2, 3), keepdims=True, dtype=np.float64)
        v = x.var(axis=(2, 3), keepdims=True, dtype=np.float64)
        return ((x - m) / np.sqrt(v + 1e-5)).astype(F32)
    s = sd[key + ".weight"] / np.sqrt(sd[key + ".running_var"] + F32(1e-5))      # eval-mode BatchNorm2d
    return ((x - sd[key + ".running_mean"][None, :, None, None]) * s[None, :, None, None] +
            sd[key + ".bias"][None, :, None, None]).astype(F32)


def basic_encoder(x, sd, p, kind):
    """BasicEncoder.forward (extractor.py:165-189) without dropout; kind = 'instance' (fnet) / 'batch' (cnet)."""
    relu = lambda t: np.maximum(t, 0)
    x = relu(_norm(_conv(x, sd, p + "conv1", 2, (3, 3)), sd, p + "norm1", kind))
    for li, stride in ((1, 1), (2, 2), (3, 2)):
        for bi, s in ((0, stride), (1, 1)):
            q = f"{p}layer{li}.{bi}."
            y = relu(_norm(_conv(x, sd, q + "conv1", s, (1, 1)), sd, q + "norm1", kind))
            y = relu(_norm(_conv(y, sd, q + "conv2", 1, (1, 1)), sd, q + "norm2", kind))
            if s != 1:   # downsample = Sequential(conv1x1 stride s, norm3); loaded from the `downsample.1` entries
                x = _norm(_conv(x, sd, q + "downsample.0", s), sd, q + "downsample.1", kind)
            x = relu(x + y)
    return _conv(x, sd, p + "conv2")


def update_block(net, inp, corr, flow, sd):
    """BasicUpdateBlock.forward (update.py:127-139): (net, mask, delta_flow)."""
    relu = lambda t: np.maximum(t, 0)
    sig = lambda t: (1.0 / (1.0 + np.exp(-t.astype(np.float64)))).astype(F32)
    u = "update_block."
    cor = relu(_conv(corr, sd, u + "encoder.convc1"))
    cor = relu(_conv(cor, sd, u + "encoder.convc2", 1, (1, 1)))
    flo = relu(_conv(flow, sd, u + "encoder.convf1", 1, (3, 3)))
    flo = relu(_conv(flo, sd, u + "encoder.convf2", 1, (1, 1)))
    out = relu(_conv(np.concatenate([cor, flo], 1), sd, u + "encoder.conv", 1, (1, 1)))
    x = np.concatenate([inp, out, flow], 1)
    h = net
    for tag, pad in (("1", (0, 2)), ("2", (2, 0))):            # SepConvGRU: (1,5) then (5,1)
        hx = np.concatenate([h, x], 1)
        z = sig(_conv(hx, sd, u + "gru.convz" + tag, 1, pad))
        r = sig(_conv(hx, sd, u + "gru.convr" + tag, 1, pad))
        q = np.tanh(_conv(np.concatenate([r * h, x], 1), sd, u + "gru.convq" + tag, 1, pad))
        h = (1 - z) * h + z * q
    delta = _conv(relu(_conv(h, sd, u + "flow_head.conv1", 1, (1, 1))), sd, u + "flow_head.conv2", 1, (1, 1))
    mask = F32(0.25) * _conv(relu(_conv(h, sd, u + "mask.0", 1, (1, 1))), sd, u + "mask.2")
    return h.astype(F32), mask.astype(F32), delta.astype(F32)


def upsample_flow(flow, mask):
    """RAFT.upsample_flow (raft.py:72-84): convex combination of the 3x3 neighbourhood of 8 * flow."""
    N, _, H, W = flow.shape
    m = mask.reshape(N, 1, 9, 8, 8, H, W).astype(np.float64)
    m = np.exp(m - m.max(axis=2, keepdims=True))
    m = m / m.sum(axis=2, keepdims=True)
    fp = np.pad(8.0 * flow.astype(np.float64), ((0, 0), (0, 0), (1, 1), (1, 1)))
    nb = np.stack([fp[:, :, ky:ky + H, kx:kx + W] for ky in range(3) for kx in range(3)], axis=2)   # unfold order
    up = (m * nb.reshape(N, 2, 9, 1, 1, H, W)).sum(axis=2)
    return up.transpose(0, 1, 4, 2, 5, 3).reshape(N, 2, 8 * H, 8 * W).astype(F32)


def raft_forward(sd, image1, image2, iters=12):
    """RAFT.forward(..., test_mode=True) (raft.py:86-144): images (N,3,H,W) in [0,255] -> (flow_low, flow_up)."""
    sd = {k: np.asarray(v, dtype=F32) for k, v in sd.items() if not k.endswith("num_batches_tracked")}
    i1 = (2 * (image1.astype(F32) / F32(255.0)) - 1).astype(F32)
    i2 = (2 * (image2.astype(F32) / F32(255.0)) - 1).astype(F32)
    N = i1.shape[0]
    f = basic_encoder(np.concatenate([i1, i2], 0), sd, "fnet.", "instance")
    fmap1, fmap2 = f[:N], f[N:]
    c = basic_encoder(i1, sd, "cnet.", "batch")
    net, inp = np.tanh(c[:, :128]), np.maximum(c[:, 128:], 0)
    H, W = fmap1.shape[2:]
    ys, xs = np.meshgrid(np.arange(H, dtype=F32), np.arange(W, dtype=F32), indexing="ij")
    coords0 = np.broadcast_to(np.stack([xs, ys], 0)[None], (N, 2, H, W)).astype(F32)
    coords1 = coords0.copy()
    mask = None
    for _ in range(iters):
        corr = alternate_corr_block(fmap1, fmap2, coords1, 4, 4)
        net, mask, delta = update_block(net, inp, corr, coords1 - coords0, sd)
        coords1 = coords1 + delta
    flow = (coords1 - coords0).astype(F32)
    return flow, upsample_flow(flow, mask)
