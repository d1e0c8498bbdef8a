#!/usr/bin/env python3
"""Authoring container only (needs /root/reference): the reference's OWN op_cpu path and the oracle, back to back, on the same
cores -- so that the bias of bench.py's `cpu_baseline` (kind "port": the oracle) against the real reference is on record
(missing 4 / next 7b).  One VToonify-D frame 22x256x256 -> 3x1024x1024, fp32, `--threads` host threads.

    python tools/cpu_port_vs_reference.py --threads 8 --reps 3 > profiles/r04_cpu_port_vs_reference.txt
"""
import argparse
import importlib
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument("--threads", type=int, default=8)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--size", type=int, default=256)
args = ap.parse_args()
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("VTOONIFY_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
import model.stylegan  # noqa: E402
_cpu = importlib.import_module("model.stylegan.op_cpu")
_gf = importlib.import_module("model.stylegan.op_cpu.conv2d_gradfix")
sys.modules["model.stylegan.op"] = _cpu            # op_cpu/readme.md:5-12, applied at run time
sys.modules["model.stylegan.op.conv2d_gradfix"] = _gf
_cpu.conv2d_gradfix = _gf
import numpy as np  # noqa: E402
import torch  # noqa: E402
from model.vtoonify import VToonify  # noqa: E402
sys.path.append(REPO)
from oracle import vtoonify_oracle as O  # noqa: E402
from vtoonify_amd import synth  # noqa: E402

torch.set_grad_enabled(False)
torch.set_num_threads(args.threads)
m = VToonify(backbone="dualstylegan").eval()
shapes = {k: list(v.shape) for k, v in m.state_dict().items()}
sd = synth.synth_state_dict(shapes, seed=0)
m.load_state_dict(sd)
x = synth.synth_frames(1, args.size, args.size, seed=2)
s = synth.synth_style(seed=17)
sdn = synth.to_numpy_sd(sd)
O.set_backend("torch")
print(f"# VToonify-D, one frame 22x{args.size}x{args.size} -> 3x{4 * args.size}x{4 * args.size}, fp32, {args.threads} host threads "
      f"(os.cpu_count() = {os.cpu_count()}), torch {torch.__version__}; alternating, {args.reps} repetitions each")
tr, to = [], []
y_ref = y_or = None
for r in range(args.reps + 1):   # first pass = warm-up of both
    t0 = time.perf_counter(); y_ref = m(x, s, d_s=0.5); t1 = time.perf_counter()
    y_or = O.vtoonify_forward(sdn, x.numpy(), s.numpy(), 0.5, "dualstylegan"); t2 = time.perf_counter()
    if r:
        tr.append(t1 - t0); to.append(t2 - t1)
        print(f"rep {r}: reference op_cpu path {t1 - t0:7.2f} s   oracle (torch contractions) {t2 - t1:7.2f} s")
err = float(np.abs(y_ref.numpy() - y_or).max() / np.abs(y_ref.numpy()).max())
mr, mo = sorted(tr)[len(tr) // 2], sorted(to)[len(to) // 2]
print(f"median: reference {mr:.2f} s/frame = {1 / mr:.3f} frames/s; oracle {mo:.2f} s/frame = {1 / mo:.3f} frames/s; "
      f"reference / oracle time = {mr / mo:.2f} (the port understates the reference's CPU rate by {mo / mr:.2f}x)")
print(f"oracle vs reference on this frame: max-rel {err:.2e}")
