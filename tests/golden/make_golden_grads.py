#!/usr/bin/env python3
"""Gradient goldens from the REAL reference (authoring container only; needs /root/reference, read-only).

The reference's StyledConv(upsample) -> StyledConv -> ToRGB graph (model/stylegan/model.py:323-392) on its op_cpu
operator twin (op_cpu/readme.md:5-12 swap, applied at run time): first-order gradients, path-length-style second
order (g_path_regularize, util.py:91-99) and R1-style second order under no_weight_gradients() (d_r1_loss,
util.py:75-82).  Stored with every parameter / input so that tests/test_grad_golden.py can rebuild the same
graph on vtoonify_amd.op WITHOUT the reference (the GPU box has none) and check the HIP backward kernels on
hardware.  Only tensors the reference computes are stored.

    python tests/golden/make_golden_grads.py      # rewrites tests/golden/grads.npz
"""
import importlib
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("VTOONIFY_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
import model.stylegan  # noqa: E402

_cpu = importlib.import_module("model.stylegan.op_cpu")
_gf = importlib.import_module("model.stylegan.op_cpu.conv2d_gradfix")
sys.modules["model.stylegan.op"] = _cpu
sys.modules["model.stylegan.op.conv2d_gradfix"] = _gf
_cpu.conv2d_gradfix = _gf

import numpy as np  # noqa: E402
import torch  # noqa: E402
from model.stylegan.model import StyledConv, ToRGB  # noqa: E402

torch.manual_seed(0)
up = StyledConv(16, 8, 3, 32, upsample=True)
same = StyledConv(8, 8, 3, 32)
rgb = ToRGB(8, 32, upsample=False)
for m in (up, same):
    torch.nn.init.normal_(m.activate.bias, std=0.1)
    torch.nn.init.normal_(m.noise.weight, std=0.1)
named = [(f"{nm}.{k}", p) for nm, m in (("up", up), ("same", same), ("rgb", rgb)) for k, p in m.named_parameters()]
params = [p for _, p in named if p.requires_grad]
x = torch.randn(2, 16, 6, 5, requires_grad=True)
s = torch.randn(2, 32, requires_grad=True)
proj = torch.randn(2, 3, 12, 10)
noise = 0.3 * torch.randn(2, 1, 12, 10)
y = up(x, s, noise=noise)
y = same(y, s, noise=noise)
img = rgb(y, s)
g1 = torch.autograd.grad((img * proj).sum(), [x, s] + params, create_graph=True, allow_unused=True)
g2 = torch.autograd.grad(g1[1].pow(2).sum(), [x] + params, retain_graph=True, allow_unused=True)
with _gf.no_weight_gradients():
    gx, = torch.autograd.grad(img.sum(), [x], create_graph=True)
g3 = torch.autograd.grad(gx.pow(2).sum(), params, allow_unused=True)

res = {"x": x, "s": s, "proj": proj, "noise": noise, "img": img, "blur_kernel": up.conv.blur.kernel}
for k, p in named:
    res["p__" + k] = p
names = ["x", "s"] + [k for k, p in named if p.requires_grad]
for tag, gs, nm in (("g1", g1, names), ("g2", g2, ["x"] + names[2:]), ("g3", g3, names[2:])):
    for k, g in zip(nm, gs):
        if g is not None:
            res[f"{tag}__{k}"] = g
out = {k: np.ascontiguousarray(v.detach().numpy()) for k, v in res.items()}
path = os.path.join(HERE, "grads.npz")
np.savez_compressed(path, **out)
print(f"wrote grads.npz: {os.path.getsize(path) / 1024:.1f} KiB, {len(out)} arrays")
print(sorted(out))
