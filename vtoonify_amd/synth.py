"""Deterministic synthetic weights and frames for parity tests and the benchmark.

There are no checkpoints in the reference tree (checkpoint/README.md lists download
locations only) and no network, so every parity / benchmark run uses synthetic
weights.  The values are a pure function of (state_dict key, shape, seed): the golden
generator (tests/golden/make_golden.py, which imports the real reference) and the
HIP path therefore load bit-identical tensors without shipping 600 MB of weights.

Scales follow the reference initialisers so activations stay O(1) through the net:
  * StyleGAN2 weights are N(0,1) with run-time equalised-lr scaling
    (model/stylegan/model.py:99-101, 140, 214-216)
  * nn.Conv2d / nn.Linear: fan-in scaled
  * the structure-transform linears are identity-like (model/dualstylegan.py:72-79)
  * ModRes conv filters are NOT shrunk by 0.01 (model/dualstylegan.py:35-36 does that
    at init) so that an error in the dilated-conv / AdaIN path cannot hide.
Buffers that carry semantics (FIR kernels) are rebuilt exactly as
model/stylegan/model.py:21-29,37,81-85 builds them.
"""
from __future__ import annotations

import math
import zlib

import numpy as np
import torch


def _gen(key: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) + 7919 * seed) & 0x7FFFFFFF)
    return g


def fir_kernel_2d(taps=(1, 3, 3, 1), gain: float = 1.0) -> torch.Tensor:
    """Separable low-pass FIR, normalised to sum 1 then multiplied by `gain`.

    Mirrors make_kernel (model/stylegan/model.py:21-29)."""
    k = torch.tensor(taps, dtype=torch.float32)
    k = k[None, :] * k[:, None]
    k = k / k.sum()
    return k * gain


def synth_tensor(key: str, shape, seed: int = 0) -> torch.Tensor:
    shape = tuple(int(s) for s in shape)
    g = _gen(key, seed)
    leaf = key.rsplit(".", 1)[-1]

    def randn(scale=1.0):
        return torch.randn(shape, generator=g, dtype=torch.float32) * scale

    # --- pSp GradualStyleEncoder (BatchNorm running stats, PReLU, SE) -------------------
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.int64)
    if leaf == "running_mean":
        return randn(0.1)
    if leaf == "running_var":
        return torch.rand(shape, generator=g, dtype=torch.float32) + 0.5
    if leaf in ("weight", "bias") and len(shape) == 1 and (".res_layer." in key or ".shortcut_layer." in key
                                                             or key.startswith("input_layer.")):
        # BatchNorm affine (weight ~ 1, bias ~ 0) and PReLU slopes (0.25 at init)
        is_prelu = key.endswith("res_layer.2.weight") or key == "input_layer.2.weight"
        if is_prelu:
            return 0.25 + randn(0.05)
        return (1.0 + randn(0.1)) if leaf == "weight" else randn(0.1)

    if key.startswith("body.") and leaf == "weight" and len(shape) == 4:
        # IR-SE-50 convs: half of He init on the residual branch and small SE weights keep the
        # activations O(10) through 24 residual units (full He init reaches 2e4 and saturates
        # every SE gate, which makes the net ill-conditioned for a parity test)
        fan_in = shape[1] * shape[2] * shape[3]
        if ".res_layer.5." in key:
            return randn(0.25 * math.sqrt(2.0 / fan_in))
        if ".res_layer." in key:
            return randn(0.5 * math.sqrt(2.0 / fan_in))

    # --- RAFT (model/raft/core): eval-mode BatchNorm affine of the context encoder ~ (1, 0) ---
    if key.startswith(("cnet.", "fnet.")) and len(shape) == 1 and leaf in ("weight", "bias") and \
            ("norm" in key or "downsample.1" in key):
        return (1.0 + randn(0.1)) if leaf == "weight" else randn(0.1)

    # --- BiSeNet face parsing (model/bisenet): eval-mode BatchNorm affine ~ (1, 0) -----------
    if key.startswith(("cp.", "ffm.", "conv_out")) and len(shape) == 1 and leaf in ("weight", "bias"):
        return (1.0 + randn(0.1)) if leaf == "weight" else randn(0.1)
    if key.startswith(("cp.", "ffm.", "conv_out")) and len(shape) == 4 and leaf == "weight":
        # 0.6 x He init keeps the activations O(10) through the 8 residual blocks; small attention
        # convs keep the sigmoid gates of ARM / FFM off saturation (a saturated gate hides errors)
        fan_in = shape[1] * shape[2] * shape[3]
        scale = 0.2 if "conv_atten" in key else 2.0 if key in ("ffm.conv1.weight", "ffm.conv2.weight") else 0.6
        return randn(scale * math.sqrt(2.0 / fan_in))

    # --- buffers with fixed semantics -------------------------------------------------
    if key.endswith("blur.kernel"):
        # ModulatedConv2d(upsample=True): Blur(kernel, upsample_factor=2) -> *4
        return fir_kernel_2d(gain=4.0)
    if key.endswith("upsample.kernel"):
        return fir_kernel_2d(gain=4.0)
    if ".noises." in key:
        return randn()
    if key.endswith("noise.weight"):
        return randn(0.1)  # harmless: inference noise is x0 (model/vtoonify.py:267)
    if key.endswith("input.input"):
        return randn()

    # --- structure transform T_s / colour path linears (identity-like) -----------------
    parts = key.split(".")
    if len(parts) == 4 and parts[0] == "generator" and parts[1] == "res" and parts[2].isdigit() \
            and int(parts[2]) >= 7:
        if leaf == "weight":
            return torch.eye(shape[0]) * (shape[0] ** 0.5) + randn(0.01) * 4.0
        return randn(0.05)

    # --- AdaIN style linears: bias = [1]*C + [0]*C (+ jitter) ---------------------------
    if ".norm.style." in key or ".norm2.style." in key:
        if leaf == "weight":
            return randn(1.0 / math.sqrt(shape[1]))
        half = shape[0] // 2
        b = randn(0.1)
        b[:half] += 1.0
        return b

    # --- equalised-lr mapping network (lr_mul = 0.01 => weight = randn / lr_mul) --------
    if ".style." in key and leaf in ("weight", "bias") and ".norm" not in key:
        if leaf == "weight":
            return randn(100.0)
        return randn(5.0)  # multiplied by lr_mul=0.01 at run time

    # --- modulation linears -------------------------------------------------------------
    if ".modulation." in key:
        if leaf == "weight":
            return randn()
        return 1.0 + randn(0.1)

    # --- StyleGAN weights (5-D modulated conv, 4-D EqualConv2d) ---------------------------
    if leaf == "weight" and len(shape) == 5:
        return randn()
    if (key.startswith("res.") or key.startswith("generator.res.")) and leaf == "weight" \
            and len(shape) == 4:
        return randn()  # EqualConv2d inside AdaResBlock, run-time scaled
    if leaf == "bias" and len(shape) == 4:
        return randn(0.1)  # ToRGB bias (1,3,1,1)

    # --- plain nn.Conv2d / nn.Linear ------------------------------------------------------
    if leaf == "weight" and len(shape) == 4:
        fan_in = shape[1] * shape[2] * shape[3]
        return randn(math.sqrt(2.0 / fan_in))
    if leaf == "weight" and len(shape) == 2:
        return randn(1.0 / math.sqrt(shape[1]))
    if leaf == "bias":
        return randn(0.1)
    if leaf == "weight":
        return randn()
    return randn()


def synth_state_dict(shapes: dict, seed: int = 0) -> dict:
    """shapes: {key: shape}.  Returns {key: fp32 CPU tensor}."""
    return {k: synth_tensor(k, s, seed) for k, s in shapes.items()}


def synth_frames(batch: int, height: int, width: int, seed: int = 1234) -> torch.Tensor:
    """(B,22,H,W) fp32: RGB in [-1,1] + 19 parsing logits / 16 (SURVEY.md 8d;
    model/vtoonify.py:162, style_transfer.py:171-174)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    rgb = torch.rand(batch, 3, height, width, generator=g) * 2 - 1
    par = torch.randn(batch, 19, height, width, generator=g) * 0.25
    return torch.cat([rgb, par], 1).contiguous()


def synth_style(seed: int = 4321, n_latent: int = 18, dim: int = 512) -> torch.Tensor:
    """(1,18,512) W+ code."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return torch.randn(1, n_latent, dim, generator=g)


def to_numpy_sd(sd: dict) -> dict:
    return {k: np.ascontiguousarray(v.detach().cpu().numpy()) for k, v in sd.items()}
