#!/usr/bin/env python
"""A/B of VToonifyEngine constructor options on the headline workload (D, 4 x 22x256x256, bf16, 3 lanes, hipGraph replay),
alternating the arms REPS times on one box:

    python tools/ab_engine.py "fuse_rgb128=True" "fuse_rgb128=False" [--reps 3] [--steps 60] [--batch 4] [--size 256]

Prints frames/s per arm and repetition, and the per-launch times (HIP events, engine.time_ops) of the same-resolution
StyledConv / ToRGB launches of each arm."""
import ast
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

import bench  # noqa: E402
from vtoonify_amd import synth  # noqa: E402
from vtoonify_amd.engine import VToonifyEngine  # noqa: E402

args = [a for a in sys.argv[1:] if "=" in a and not a.startswith("--")]


def opt(name, default):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


reps, steps, B, S = opt("--reps", 3), opt("--steps", 60), opt("--batch", 4), opt("--size", 256)
NL = opt("--lanes", 3)
dev = torch.device("cuda:0")
sd = {k: v.to(dev) for k, v in synth.synth_state_dict(bench.state_shapes("dualstylegan"), 0).items()}
style = synth.synth_style(seed=17).to(dev)
pool = [synth.synth_frames(B, S, S, seed=i).to(dev) for i in range(4)]
lanes = NL
streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(lanes - 1)]
arms = []
envs = {}


class arm_env:   # the arm's environment switches, for the duration of a with-block
    def __init__(self, name):
        self.kv = envs.get(name, {})

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


for a in args:
    kw = {}
    for kv in a.split(","):
        k, v = kv.split("=")
        if k == "hints":   # hints=FILE.json: a tile-hint table {conv signature: tile_hint} (engine.conv_signature)
            import json
            kw["tile_hints"] = {kk: int(vv) for kk, vv in json.load(open(v)).items()}
        elif k == "env":   # env=NAME:VALUE: an environment switch of the library, set while this arm builds plans, captures and runs
            envs.setdefault(a, {})[v.split(":")[0]] = v.split(":")[1]
        elif k == "lib":   # lib=PATH: an experiment build of the library (vtoonify_amd.build --variant); an engine keeps the
            libpath = v    # handle it was constructed with, so arms with different libraries coexist in one process
        else:
            kw[k] = ast.literal_eval(v)
    from vtoonify_amd import _lib
    _lib.use_library(locals().pop("libpath", None) or _lib.DEFAULT_LIB)
    arms.append((a, VToonifyEngine(sd, "dualstylegan", 256, torch.bfloat16, dev, **kw)))
    _lib.use_library(_lib.DEFAULT_LIB)


def rate(eng):
    def step(i):
        ln = i % lanes
        with torch.cuda.stream(streams[ln]):
            return eng.forward(pool[i % 4], style, 0.5, shared_style=True, use_graph=True, lane=ln, borrow=True)
    for i in range(lanes + 3):
        step(i)
        if i < lanes:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    return steps * B / (time.perf_counter() - t0)


outs = []
for name, eng in arms:
    with arm_env(name):
        outs.append(eng.forward(pool[0], style, 0.5, shared_style=True, use_graph=False, lane=0).clone())
print("outputs equal across arms:", all(torch.equal(outs[0], o) for o in outs[1:]),
      "max abs diff", max(float((outs[0] - o).abs().max()) for o in outs[1:]) if len(outs) > 1 else 0.0)
for r in range(reps):
    for name, eng in arms:
        with arm_env(name):
            print(f"rep {r}  {name:<40} {rate(eng):8.1f} frames/s", flush=True)
for name, eng in arms:
    with arm_env(name):
        plan = eng.plan_for(B, S, S, True, True)
        per = eng.time_ops(plan, iters=5)
    tot = sum(ms for _, ms in per)
    print(f"--- {name}: {len(per)} launches, kernel sum {tot:.3f} ms")
    for info, ms in per:
        if info.get("name") == "conv" and ("--all" in sys.argv or ((info["cout"] == 3 or info.get("hw", (0, 0))[0] >= 64) and
                                                            ":up" not in info["sig"] and info["cin"] <= 512)):
            print(f"   {info['sig']:<34} {info['kernel']:<34} {1e3 * ms:8.1f} us")
