#!/usr/bin/env python3
"""The reference's OWN op_cpu path timed on the host cores (bench.py's `cpu_baseline` with kind "reference"): one VToonify frame
through /root/reference's model/vtoonify.py:210-277 over model/stylegan/op_cpu, seeded synthetic weights.  Runs in its own process
(it aliases model.stylegan.op to op_cpu as op_cpu/readme.md:5-12 prescribes) and only where the reference is mounted
(VTOONIFY_REFERENCE or /root/reference: the authoring container; the GPU boxes do not have it).  Prints one JSON line.

    python tools/cpu_reference.py --threads 32 --height 256 --width 256 --budget 25
"""
import argparse
import importlib
import json
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument("--threads", type=int, default=8)
ap.add_argument("--height", type=int, default=256)
ap.add_argument("--width", type=int, default=256)
ap.add_argument("--backbone", default="dualstylegan")
ap.add_argument("--budget", type=float, default=25.0)
args = ap.parse_args()
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("VTOONIFY_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
import model.stylegan  # noqa: E402,F401
_cpu = importlib.import_module("model.stylegan.op_cpu")
_gf = importlib.import_module("model.stylegan.op_cpu.conv2d_gradfix")
sys.modules["model.stylegan.op"] = _cpu
sys.modules["model.stylegan.op.conv2d_gradfix"] = _gf
_cpu.conv2d_gradfix = _gf
import torch  # noqa: E402
from model.vtoonify import VToonify  # noqa: E402
sys.path.append(REPO)
from vtoonify_amd import synth  # noqa: E402

torch.set_grad_enabled(False)
torch.set_num_threads(args.threads)
m = VToonify(backbone=args.backbone).eval()
m.load_state_dict(synth.synth_state_dict({k: list(v.shape) for k, v in m.state_dict().items()}, seed=0))
s = synth.synth_style(seed=17)
d_s = 0.5 if args.backbone == "dualstylegan" else None
xp = synth.synth_frames(1, 64, 64, seed=1)
m(xp, s, d_s=d_s)
t0 = time.perf_counter()
m(xp, s, d_s=d_s)
t_probe = time.perf_counter() - t0
h, w = args.height, args.width
while t_probe * (h * w) / (64 * 64) > args.budget and h * w > 64 * 64:
    h, w = max(h // 2, 64), max(w // 2, 64)
reps = [t_probe]
if (h, w) != (64, 64):
    x = synth.synth_frames(1, h, w, seed=2)
    reps, t_all = [], time.perf_counter()
    while len(reps) < 3 and (not reps or time.perf_counter() - t_all + reps[-1] < args.budget):
        t0 = time.perf_counter()
        y = m(x, s, d_s=d_s)
        reps.append(time.perf_counter() - t0)
    assert bool(torch.isfinite(y).all())
print(json.dumps({"h": h, "w": w, "reps_s": reps, "threads": args.threads, "reference": REF}))
