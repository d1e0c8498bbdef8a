"""CPU oracle for the pSp style encoder (GradualStyleEncoder).  TEST INFRASTRUCTURE ONLY.

numpy fp32 restatement of model/encoder/encoders/psp_encoders.py:11-116 and
model/encoder/encoders/helpers.py:53-119 of the reference (IR-SE-50 trunk, FPN, 18 map2style
heads), eval mode.  Pinned against tensors computed by the reference itself
(tests/golden/make_golden_psp.py -> tests/golden/psp.npz; tests/test_oracle_golden.py).
Third-party arithmetic restated from its published definition: nn.BatchNorm2d (eval:
(x - running_mean) / sqrt(running_var + 1e-5) * weight + bias), nn.PReLU (per-channel slope),
nn.LeakyReLU() (slope 0.01), nn.AdaptiveAvgPool2d(1), nn.MaxPool2d(1, stride) (= subsampling),
F.interpolate(bilinear, align_corners=True).  Convolutions / linears come from
oracle.vtoonify_oracle (numpy or torch backend).
"""
from __future__ import annotations

import numpy as np

from . import vtoonify_oracle as O

F32 = np.float32
BLOCKS_50 = [(64, 64, 3), (64, 128, 4), (128, 256, 14), (256, 512, 3)]  # helpers.py:33-39


def units(num_layers=50):
    """[(in_channel, depth, stride)] of the 24 bottlenecks (helpers.py:28-30, 32-39)."""
    assert num_layers == 50
    out = []
    for cin, depth, n in BLOCKS_50:
        out.append((cin, depth, 2))
        out += [(depth, depth, 1)] * (n - 1)
    return out


def batch_norm(sd, prefix, x):
    w, b = sd[prefix + "weight"], sd[prefix + "bias"]
    m, v = sd[prefix + "running_mean"], sd[prefix + "running_var"]
    scale = (w / np.sqrt(v + F32(1e-5))).astype(F32)
    return (x * scale.reshape(1, -1, 1, 1) + (b - m * scale).reshape(1, -1, 1, 1)).astype(F32)


def prelu(x, slope):
    return np.where(x >= 0, x, x * slope.reshape(1, -1, 1, 1)).astype(F32)


def se_module(sd, prefix, x):
    """SEModule (helpers.py:53-69): x * sigmoid(fc2(relu(fc1(avgpool(x)))))."""
    s = x.mean(axis=(2, 3), keepdims=True, dtype=np.float64).astype(F32)
    s = np.maximum(O.conv2d(s, sd[prefix + "fc1.weight"]), 0)
    s = O.conv2d(s, sd[prefix + "fc2.weight"])
    return (x * (1.0 / (1.0 + np.exp(-s.astype(np.float64)))).astype(F32)).astype(F32)


def bottleneck_ir_se(sd, prefix, x, cin, depth, stride):
    """bottleneck_IR_SE (helpers.py:97-119)."""
    if cin == depth:
        shortcut = x[:, :, ::stride, ::stride]                       # MaxPool2d(1, stride)
    else:
        shortcut = batch_norm(sd, prefix + "shortcut_layer.1.",
                              O.conv2d(x, sd[prefix + "shortcut_layer.0.weight"], stride=stride))
    r = batch_norm(sd, prefix + "res_layer.0.", x)
    r = prelu(O.conv2d(r, sd[prefix + "res_layer.1.weight"], padding=1), sd[prefix + "res_layer.2.weight"])
    r = batch_norm(sd, prefix + "res_layer.4.", O.conv2d(r, sd[prefix + "res_layer.3.weight"], stride=stride,
                                                         padding=1))
    r = se_module(sd, prefix + "res_layer.5.", r)
    return (r + shortcut).astype(F32)


def upsample_add(x, y):
    """_upsample_add (psp_encoders.py:71-88): bilinear, align_corners=True, to y's size, + y."""
    n, c, h, w = x.shape
    H, W = y.shape[2], y.shape[3]
    ys = (np.arange(H, dtype=F32) * (F32(h - 1) / F32(H - 1) if H > 1 else F32(0))).astype(F32)
    xs = (np.arange(W, dtype=F32) * (F32(w - 1) / F32(W - 1) if W > 1 else F32(0))).astype(F32)
    y0 = np.minimum(ys.astype(np.int64), h - 1)
    x0 = np.minimum(xs.astype(np.int64), w - 1)
    y1, x1 = np.minimum(y0 + 1, h - 1), np.minimum(x0 + 1, w - 1)
    ly, lx = (ys - y0).astype(F32).reshape(1, 1, H, 1), (xs - x0).astype(F32).reshape(1, 1, 1, W)
    a, b = x[:, :, y0][:, :, :, x0], x[:, :, y0][:, :, :, x1]
    c_, d = x[:, :, y1][:, :, :, x0], x[:, :, y1][:, :, :, x1]
    top, bot = a + (b - a) * lx, c_ + (d - c_) * lx
    return (top + (bot - top) * ly + y).astype(F32)


def style_block(sd, prefix, x, spatial):
    """GradualStyleBlock (psp_encoders.py:11-32): log2(spatial) x [conv s2 + LeakyReLU(0.01)], EqualLinear."""
    n = int(np.log2(spatial))
    for i in range(n):
        x = O.leaky_relu(O.conv2d(x, sd[f"{prefix}convs.{2 * i}.weight"], sd[f"{prefix}convs.{2 * i}.bias"],
                                  stride=2, padding=1), 0.01)
    x = x.reshape(-1, x.shape[1])
    return O.equal_linear(x, sd[prefix + "linear.weight"], sd[prefix + "linear.bias"], 1.0, False)


def gradual_style_encoder(sd, x, n_styles=18, return_taps=False):
    """GradualStyleEncoder.forward (psp_encoders.py:90-116).  x: (B, 3, H, W) -> (B, n_styles, 512)."""
    x = np.asarray(x, dtype=F32)
    x = prelu(batch_norm(sd, "input_layer.1.", O.conv2d(x, sd["input_layer.0.weight"], padding=1)),
              sd["input_layer.2.weight"])
    taps = {}
    for i, (cin, depth, stride) in enumerate(units()):
        x = bottleneck_ir_se(sd, f"body.{i}.", x, cin, depth, stride)
        if i in (6, 20, 23):
            taps[i] = x
    c1, c2, c3 = taps[6], taps[20], taps[23]
    lat = [style_block(sd, f"styles.{j}.", c3, 16) for j in range(3)]
    p2 = upsample_add(c3, O.conv2d(c2, sd["latlayer1.weight"], sd["latlayer1.bias"]))
    lat += [style_block(sd, f"styles.{j}.", p2, 32) for j in range(3, 7)]
    p1 = upsample_add(p2, O.conv2d(c1, sd["latlayer2.weight"], sd["latlayer2.bias"]))
    lat += [style_block(sd, f"styles.{j}.", p1, 64) for j in range(7, n_styles)]
    out = np.stack(lat, axis=1).astype(F32)
    if return_taps:
        return out, (c1, c2, c3, p2, p1)
    return out
