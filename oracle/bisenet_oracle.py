"""CPU oracle for the BiSeNet face-parsing network.  TEST INFRASTRUCTURE ONLY.

numpy fp32 restatement of the reference's model/bisenet/model.py:13-254 (ConvBNReLU,
BiSeNetOutput, AttentionRefinementModule, ContextPath, FeatureFusionModule, BiSeNet) and
model/bisenet/resnet.py:14-80 (BasicBlock, Resnet18), eval mode, plus the parsing pre/post
processing of the video loop (style_transfer.py:171-172).  Pinned against tensors computed by the
reference itself (tests/golden/make_golden_bisenet.py -> tests/golden/bisenet.npz;
tests/test_oracle_golden.py).  Third-party arithmetic restated from its published definition:
nn.BatchNorm2d (eval), nn.MaxPool2d(3, 2, 1) (-inf padding), F.avg_pool2d over the whole map,
F.interpolate nearest (src = floor(dst * in/out)) and bilinear (aten
area_pixel_compute_source_index, both corner conventions).  Convolutions come from
oracle.vtoonify_oracle (numpy or torch backend).  Only tests/, __graft_entry__.smoke() and
bench tools' cpu_baseline legs may import this module.
"""
from __future__ import annotations

import numpy as np

from . import vtoonify_oracle as O
from .psp_oracle import batch_norm

F32 = np.float32


def relu(x):
    return np.maximum(x, F32(0)).astype(F32)


def sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(F32)


def conv_bn_relu(sd, prefix, x, stride=1, padding=1):
    """ConvBNReLU (model.py:13-29): conv (no bias) -> BatchNorm -> ReLU."""
    return relu(batch_norm(sd, prefix + "bn.", O.conv2d(x, sd[prefix + "conv.weight"], stride=stride,
                                                       padding=padding)))


def max_pool_3x3_s2(x):
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1) (resnet.py:63)."""
    n, c, h, w = x.shape
    oh, ow = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    xp = np.full((n, c, h + 2, w + 2), -np.inf, dtype=F32)
    xp[:, :, 1:h + 1, 1:w + 1] = x
    out = np.full((n, c, oh, ow), -np.inf, dtype=F32)
    for ky in range(3):
        for kx in range(3):
            out = np.maximum(out, xp[:, :, ky:ky + 2 * oh - 1:2, kx:kx + 2 * ow - 1:2])
    return out


def basic_block(sd, prefix, x, stride):
    """BasicBlock.forward (resnet.py:36-48)."""
    r = relu(batch_norm(sd, prefix + "bn1.", O.conv2d(x, sd[prefix + "conv1.weight"], stride=stride, padding=1)))
    r = batch_norm(sd, prefix + "bn2.", O.conv2d(r, sd[prefix + "conv2.weight"], padding=1))
    sc = x
    if prefix + "downsample.0.weight" in sd:
        sc = batch_norm(sd, prefix + "downsample.1.", O.conv2d(x, sd[prefix + "downsample.0.weight"], stride=stride))
    return relu(sc + r)


def resnet18(sd, prefix, x):
    """Resnet18.forward (resnet.py:68-77): feat8, feat16, feat32."""
    x = relu(batch_norm(sd, prefix + "bn1.", O.conv2d(x, sd[prefix + "conv1.weight"], stride=2, padding=3)))
    x = max_pool_3x3_s2(x)
    feats = []
    for li, stride in ((1, 1), (2, 2), (3, 2), (4, 2)):
        x = basic_block(sd, f"{prefix}layer{li}.0.", x, stride)
        x = basic_block(sd, f"{prefix}layer{li}.1.", x, 1)
        feats.append(x)
    return feats[1], feats[2], feats[3]


def global_avg(x):
    return x.mean(axis=(2, 3), keepdims=True, dtype=np.float64).astype(F32)


def interpolate_nearest(x, size):
    """F.interpolate(x, size, mode='nearest'): src = min(floor(dst * in/out), in - 1) in fp32."""
    n, c, h, w = x.shape
    H, W = size
    ys = np.minimum(np.floor(np.arange(H, dtype=F32) * (F32(h) / F32(H))).astype(np.int64), h - 1)
    xs = np.minimum(np.floor(np.arange(W, dtype=F32) * (F32(w) / F32(W))).astype(np.int64), w - 1)
    return np.ascontiguousarray(x[:, :, ys][:, :, :, xs])


def _bilinear_axis(out_size, in_size, align_corners, scale=None):
    d = np.arange(out_size, dtype=F32)
    if align_corners:
        s = F32(in_size - 1) / F32(out_size - 1) if out_size > 1 else F32(0)
        src = (s * d).astype(F32)
    else:
        s = F32(in_size) / F32(out_size) if scale is None else F32(scale)
        src = np.maximum((s * (d + F32(0.5)) - F32(0.5)).astype(F32), F32(0))
    i0 = np.minimum(src.astype(np.int64), in_size - 1)
    i1 = i0 + (i0 < in_size - 1)
    l1 = (src - i0.astype(F32)).astype(F32)
    return i0, i1, (F32(1) - l1).astype(F32), l1


def interpolate_bilinear(x, size, align_corners):
    """F.interpolate(x, size, mode='bilinear', align_corners=...) (aten upsample_bilinear2d)."""
    n, c, h, w = x.shape
    H, W = size
    y0, y1, ly0, ly1 = _bilinear_axis(H, h, align_corners)
    x0, x1, lx0, lx1 = _bilinear_axis(W, w, align_corners)
    ly0, ly1 = ly0.reshape(1, 1, H, 1), ly1.reshape(1, 1, H, 1)
    lx0, lx1 = lx0.reshape(1, 1, 1, W), lx1.reshape(1, 1, 1, W)
    a, b = x[:, :, y0][:, :, :, x0], x[:, :, y0][:, :, :, x1]
    c_, d = x[:, :, y1][:, :, :, x0], x[:, :, y1][:, :, :, x1]
    return (ly0 * (lx0 * a + lx1 * b) + ly1 * (lx0 * c_ + lx1 * d)).astype(F32)


def arm(sd, prefix, x):
    """AttentionRefinementModule.forward (model.py:78-85)."""
    feat = conv_bn_relu(sd, prefix + "conv.", x)
    att = O.conv2d(global_avg(feat), sd[prefix + "conv_atten.weight"])
    att = sigmoid(batch_norm(sd, prefix + "bn_atten.", att))
    return (feat * att).astype(F32)


def context_path(sd, prefix, x):
    """ContextPath.forward (model.py:108-129): feat8, feat16_up (x8), feat32_up (x16)."""
    feat8, feat16, feat32 = resnet18(sd, prefix + "resnet.", x)
    avg = conv_bn_relu(sd, prefix + "conv_avg.", global_avg(feat32), padding=0)
    avg_up = interpolate_nearest(avg, feat32.shape[2:])
    feat32_sum = (arm(sd, prefix + "arm32.", feat32) + avg_up).astype(F32)
    feat32_up = conv_bn_relu(sd, prefix + "conv_head32.", interpolate_nearest(feat32_sum, feat16.shape[2:]))
    feat16_sum = (arm(sd, prefix + "arm16.", feat16) + feat32_up).astype(F32)
    feat16_up = conv_bn_relu(sd, prefix + "conv_head16.", interpolate_nearest(feat16_sum, feat8.shape[2:]))
    return feat8, feat16_up, feat32_up


def feature_fusion(sd, prefix, fsp, fcp):
    """FeatureFusionModule.forward (model.py:197-208)."""
    feat = conv_bn_relu(sd, prefix + "convblk.", np.concatenate([fsp, fcp], axis=1), padding=0)
    att = relu(O.conv2d(global_avg(feat), sd[prefix + "conv1.weight"]))
    att = sigmoid(O.conv2d(att, sd[prefix + "conv2.weight"]))
    return (feat * att + feat).astype(F32)


def bisenet_output(sd, prefix, x):
    """BiSeNetOutput.forward (model.py:43-46)."""
    return O.conv2d(conv_bn_relu(sd, prefix + "conv.", x), sd[prefix + "conv_out.weight"])


def bisenet_forward(sd, x, return_taps=False):
    """BiSeNet.forward (model.py:241-254): (feat_out, feat_out16, feat_out32), each (B,19,H,W)."""
    x = np.asarray(x, dtype=F32)
    size = x.shape[2:]
    feat_res8, feat_cp8, feat_cp16 = context_path(sd, "cp.", x)
    feat_fuse = feature_fusion(sd, "ffm.", feat_res8, feat_cp8)
    outs = (bisenet_output(sd, "conv_out.", feat_fuse), bisenet_output(sd, "conv_out16.", feat_cp8),
            bisenet_output(sd, "conv_out32.", feat_cp16))
    outs = tuple(interpolate_bilinear(o, size, True) for o in outs)
    if return_taps:
        return outs, (feat_res8, feat_cp8, feat_cp16)
    return outs


def parsing_maps(sd, x):
    """style_transfer.py:171-172: x_p = nearest_x0.5(BiSeNet(2 * bilinear_x2(x))[0]) for frames x in
    [-1, 1] (B,3,H,W) -> (B,19,H,W).  (The /16 of style_transfer.py:174 is applied by the caller.)"""
    x = np.asarray(x, dtype=F32)
    n, c, h, w = x.shape
    up = interpolate_bilinear(x, (2 * h, 2 * w), False)
    y = bisenet_forward(sd, (F32(2) * up).astype(F32))[0]
    return np.ascontiguousarray(y[:, :, ::2, ::2])   # nearest, scale_factor=0.5: src = 2 * dst
