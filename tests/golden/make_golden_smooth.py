#!/usr/bin/env python3
"""Golden fixture for the parsing-map temporal smoothing from the REAL reference (authoring container only).

    python tests/golden/make_golden_smooth.py     # writes tests/golden/smooth.npz

smooth_parsing_map.py cannot be imported here (cv2 / torchvision are absent, and `warp` calls .cuda()), so this
script EXECUTES the reference's own source lines, read from /root/reference at run time: the `warp` function
(smooth_parsing_map.py:37-75) and the fusion statements of the main loop (:155-166), with `Tensor.cuda` made a
no-op and the optical flow `flow_up` supplied (RAFT is not run).  `Downsample` is the reference's
model.stylegan.model.Downsample on the op_cpu ops (the shim of tests/golden/make_golden.py).  Nothing is copied
from the reference: only tensors it computes.
"""
import importlib
import os
import sys
import textwrap

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("VTOONIFY_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch import nn  # noqa: E402

_cpu = importlib.import_module("model.stylegan.op_cpu")
_gf = importlib.import_module("model.stylegan.op_cpu.conv2d_gradfix")
sys.modules["model.stylegan.op"] = _cpu
sys.modules["model.stylegan.op.conv2d_gradfix"] = _gf
_cpu.conv2d_gradfix = _gf
from model.stylegan.model import Downsample  # noqa: E402

torch.set_grad_enabled(False)
torch.Tensor.cuda = lambda self, *a, **k: self     # `warp` moves its grid / mask to the GPU: stay on the host
SRC = os.path.join(REF, "smooth_parsing_map.py")


def lines(a, b):
    with open(SRC) as f:
        return textwrap.dedent("".join(f.readlines()[a - 1:b]))


ns = {"torch": torch, "nn": nn}
exec(compile(lines(37, 75), SRC, "exec"), ns)       # def warp(x, flo)
warp = ns["warp"]
FUSE = compile(lines(155, 166), SRC, "exec")        # output, mask = warp(...) ... parse += [down(fused_Ps)...]


def smooth_clip(g, T, H, W, CP, window, big):
    Is = torch.tanh(torch.randn(T, 3, H, W, generator=g).cumsum(0) * 0.4)          # slowly changing frames
    Ps = torch.randn(T, CP, H, W, generator=g) * 4
    flows = torch.randn(T, 2 * window + 1, 2, H, W, generator=g) * big
    flows[:, :, :, 0, 0] = torch.tensor([-3.7, 0.4])          # samples that leave the image
    flows[:, :, :, H - 1, W - 1] = torch.tensor([0.8, 1.0])   # exactly on / just over the border
    flows[:, :, :, 2, 3] = torch.tensor([1.0, -2.0])          # integer flow: zero fractional part
    Is_ = torch.cat((Is[0:window], Is, Is[-window:]), dim=0)
    Ps_ = torch.cat((Ps[0:window], Ps, Ps[-window:]), dim=0)
    wt = torch.exp(-(torch.arange(2 * window + 1).float() - window) ** 2 / (2 * ((window + 0.5) ** 2))).reshape(
        2 * window + 1, 1, 1, 1)
    down = Downsample(kernel=[1, 3, 3, 1], factor=2).eval()
    parse, fused = [], []
    for ii in range(T):
        i = ii + window
        env = {"torch": torch, "warp": warp, "window": window, "i": i, "device": "cpu", "Ps_": Ps_, "wt": wt,
               "image2": Is_[i - window:i + window + 1], "image1": Is_[i].repeat(2 * window + 1, 1, 1, 1),
               "flow_up": flows[ii], "parse": [], "down": lambda t: (fused.append(t.clone()), down(t))[1]}
        exec(FUSE, env)
        parse += env["parse"]
    return {"Is": Is.numpy(), "Ps": Ps.numpy(), "flows": flows.numpy(), "wt": wt.reshape(-1).numpy(),
            "fused": torch.cat(fused, 0).numpy(), "parse": torch.cat(parse, 0).numpy(), "cfg": np.array([window])}


def main():
    g = torch.Generator().manual_seed(11)
    out = {}
    # warp alone: (output * mask, mask)
    x = torch.randn(2, 5, 9, 14, generator=g)
    flo = torch.randn(2, 2, 9, 14, generator=g) * 2.5
    flo[:, :, 0, 0] = torch.tensor([-1.5, -0.2])
    flo[:, :, 8, 13] = torch.tensor([0.00005, 0.0])      # weights sum to 0.99995: still inside
    flo[:, :, 8, 12] = torch.tensor([1.0002, 0.0])       # 0.9998: masked
    o, m = warp(x, flo)
    out.update({"w__x": x.numpy(), "w__flo": flo.numpy(), "w__out": o.numpy(), "w__mask": m.numpy()})
    for name, (T, H, W, CP, window, big) in {"a": (4, 20, 28, 19, 2, 1.5), "b": (3, 12, 10, 19, 1, 4.0)}.items():
        for k, v in smooth_clip(g, T, H, W, CP, window, big).items():
            out[f"{name}__{k}"] = v
        print(name, out[f"{name}__parse"].shape, float(np.abs(out[f"{name}__parse"]).max()))
    path = os.path.join(HERE, "smooth.npz")
    np.savez_compressed(path, **out)
    print(f"wrote smooth.npz: {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
