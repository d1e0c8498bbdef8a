#!/usr/bin/env python3
"""Golden fixtures for the pSp style encoder from the REAL reference (authoring container only).

    python tests/golden/make_golden_psp.py     # writes tests/golden/psp.npz, keys_psp.json

GradualStyleEncoder(50, 'ir_se', opts) is imported from /root/reference (with the op -> op_cpu
alias of SURVEY.md 8c, needed by its EqualLinear import) and run in eval mode on seeded inputs with
the deterministic synthetic weights of vtoonify_amd.synth (267 M parameters are NOT stored: both
sides regenerate them from the key names).  Nothing is copied from the reference: only tensors it
computes.
"""
import argparse
import importlib
import json
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
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
from model.encoder.encoders.psp_encoders import GradualStyleEncoder  # noqa: E402

sys.path.append(REPO)
from vtoonify_amd import synth  # noqa: E402

torch.set_grad_enabled(False)


def main():
    opts = argparse.Namespace(input_nc=3, n_styles=18)
    enc = GradualStyleEncoder(50, "ir_se", opts).eval()
    shapes = {k: tuple(v.shape) for k, v in enc.state_dict().items()}
    with open(os.path.join(HERE, "keys_psp.json"), "w") as f:
        json.dump({k: list(v) for k, v in shapes.items()}, f, indent=0)
    enc.load_state_dict(synth.synth_state_dict(shapes, 0))
    g = torch.Generator().manual_seed(77)
    out = {}
    # small inputs keep the CPU-emulation tests fast (the heads still reach 1x1: stride-2 convs
    # of a 1x1 map stay 1x1); the full 256x256 case is checked on the GPU against the oracle.
    for name, (b, h, w) in {"s32": (2, 32, 32), "s64": (1, 64, 64)}.items():
        x = torch.rand(b, 3, h, w, generator=g) * 2 - 1
        feats = {}
        body = list(enc.body._modules.values())
        t = enc.input_layer(x)
        for i, l in enumerate(body):
            t = l(t)
            if i in (0, 6, 20, 23):
                feats[i] = t.clone()
        y = enc(x)
        out[name + "__x"] = x.numpy()
        out[name + "__y"] = y.numpy()
        out[name + "__c1"] = feats[6].numpy()
        out[name + "__c3"] = feats[23].numpy()
        out[name + "__b0"] = feats[0].numpy()
        print(name, tuple(y.shape), float(y.abs().max()), float(feats[23].abs().max()))
    path = os.path.join(HERE, "psp.npz")
    np.savez_compressed(path, **out)
    print(f"wrote psp.npz: {os.path.getsize(path) / 1024:.1f} KiB; {len(shapes)} state_dict keys")


if __name__ == "__main__":
    main()
