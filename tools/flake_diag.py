#!/usr/bin/env python
"""Two engine lanes in steady state (tools/flake_lanes.py) with a full diagnosis of EVERY wrong step, not only the first:

    python tools/flake_diag.py [T|D] B H W STEPS [graph|eager] [OUT.json]

Before the two-lane loop every input runs serially and the tool keeps, per input and synthesis level, the image before the
fused ToRGB ran (the up-sampled skip), the image after it, the level's activation and the eight 8-channel partial dot products
of the ToRGB (the groups one lane group of one wave holds in the conv epilogue).  A wrong step is then explained without
another GPU run: which plan buffers differ, which (image, plane, row, column range) of the first differing image, and which
candidate the wrong values equal -- the skip alone (the conv's sum never arrived), the previous frame's image or skip (a stale
read), the right value minus one lane group's / one wave's partial sum (a lost exchange), ...  Raw chunks go to OUT.json.
FLAKE_LIB = another build of the library (experiments)."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import torch  # noqa: E402

from vtoonify_amd import synth  # noqa: E402
from vtoonify_amd.engine import VToonifyEngine  # noqa: E402

bb, B, H, W, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
use_graph = (sys.argv[6] != "eager") if len(sys.argv) > 6 else True
out_json = sys.argv[7] if len(sys.argv) > 7 else None
EMU = os.environ.get("FLAKE_EMU") == "1"   # host emulation (CPU container): checks this script, cannot show the defect
dev = torch.device("cpu" if EMU else "cuda:0")
from vtoonify_amd import _lib  # noqa: E402
if EMU:
    from emu import build_emu
    _lib.use_library(build_emu.build())
elif os.environ.get("FLAKE_LIB"):
    _lib.use_library(os.environ["FLAKE_LIB"])


def sync():
    if not EMU:
        torch.cuda.synchronize()
backbone = "toonify" if bb == "T" else "dualstylegan"
from conftest import load_keys  # noqa: E402
sd = synth.synth_state_dict(load_keys(bb), 0)
eng = VToonifyEngine({k: v.to(dev) for k, v in sd.items()}, backbone, 256, torch.bfloat16, dev)
style = synth.synth_style(seed=5).to(dev)
g = torch.Generator().manual_seed(1)
NIN = 6
xs = [torch.randn(B, 22, H, W, generator=g).to(dev) for _ in range(NIN)]
d_s = 0.6 if bb == "D" else None


SPLIT = os.environ.get("FLAKE_SPLIT") == "1"   # experiment: the fused ToRGB writes the image to its OWN buffer (not in place)
if SPLIT:
    import ctypes as C
    _build = eng._build_plan

    def _build_split(*a, **k):
        plan = _build(*a, **k)
        rgbs = {t.data_ptr(): n for n, t in plan.bufs.items() if isinstance(t, torch.Tensor) and n.startswith("rgb")}
        for d, info, _, _ in plan.convs:
            if not d.rgb_out or d.rgb_out not in rgbs:
                continue
            name = rgbs[d.rgb_out]
            old = plan.bufs[name]
            new = torch.zeros_like(old)
            plan.bufs["rgbo" + name[3:]] = new
            seen = False
            for fn, args, what in plan.gen_ops:
                if what is info:
                    seen = True
                    continue
                if seen:
                    for a_ in args:
                        if isinstance(a_, C.c_void_p) and a_.value == old.data_ptr():
                            a_.value = new.data_ptr()
            d.rgb_out = new.data_ptr()
            if plan.image is old:
                plan.image = new
        return plan
    eng._build_plan = _build_split


def final_name(plan_or_snap, lvl):
    return f"rgbo{lvl}" if f"rgbo{lvl}" in (plan_or_snap.bufs if hasattr(plan_or_snap, "bufs") else plan_or_snap) else f"rgb{lvl}"


def lane_plan(ln):
    return [p for k, p in eng._plans.items() if k[-1] == ln and k[0] == B][0]


def fused_levels(plan):
    """{id(info): (level, rgb buffer name)} of the convs that carry a fused ToRGB"""
    out = {}
    ptr = {t.data_ptr(): n for n, t in plan.bufs.items() if isinstance(t, torch.Tensor) and n.startswith("rgb")}
    for d, info, _, _ in plan.convs:
        if d.rgb_out:
            out[id(info)] = ptr.get(d.rgb_resid)
    return out


# ---- serial references: final output, and per level pre / post image, activation, partial sums -----------------
ref_out, ref = [], []
for j in range(NIN):
    y = eng.forward(xs[j], style, d_s, shared_style=True, use_graph=False, lane=1).clone()
    sync()
    plan = lane_plan(1)
    fused = fused_levels(plan)
    stream = eng._stream()
    eng._run(plan.style_ops, stream)
    eng._run(plan.enc_ops, stream, plan)
    snap = {}
    for op in plan.gen_ops:
        name = fused.get(id(op[2])) if isinstance(op[2], dict) else None
        if name:
            sync()
            snap["pre." + name] = plan.bufs[name].clone()
        eng._run([op], stream, plan)
    sync()
    for n, t in plan.bufs.items():
        if isinstance(t, torch.Tensor) and (n.startswith("rgb") or n.startswith("gout") or n.startswith("up")
                                            or n.startswith("fout") or n.startswith("mask") or n.startswith("fskip")):
            snap[n] = t.clone()
    for lvl in range(5):
        wm = plan.modw[f"to_rgbs.{3 + lvl}"][0].float()          # (3, 1, C)
        act = plan.bufs[f"gout{lvl}"].float()                       # (B, h, w, C)
        C_ = act.shape[-1]
        gsz = 8
        parts = []
        for g0 in range(0, C_, gsz):
            parts.append(torch.einsum("bhwc,jc->bjhw", act[..., g0:g0 + gsz], wm[:, 0, g0:g0 + gsz]))
        snap[f"part{lvl}"] = torch.stack(parts, 0)                  # (C/8, B, 3, h, w)
    if not torch.equal(y, plan.image):
        print("WARNING: stepwise serial run differs from forward() for input", j)
    ref_out.append(y)
    ref.append(snap)
plan1 = lane_plan(1)
print("fused ToRGB levels:", sorted(v for v in fused_levels(plan1).values() if v))
for d, info, _, _ in plan1.convs:
    if d.rgb_out or info["cout"] == 3:
        print("  ", info["sig"], info["kernel"], "fused" if d.rgb_out else "")

import contextlib  # noqa: E402
streams = [None, None] if EMU else [torch.cuda.Stream(), torch.cuda.Stream()]
outs = [None, None]
hist = [[], []]   # inputs each lane has processed, in order
bad = 0
records = []
MAXDIAG = int(os.environ.get("FLAKE_MAXDIAG", "60"))


def diagnose(it, ln, j):
    sync()
    plan = lane_plan(ln + 1)
    prev = hist[ln][-2] if len(hist[ln]) >= 2 else None
    other = hist[1 - ln][-1] if hist[1 - ln] else None
    differs = []
    for n, t in plan.bufs.items():
        if isinstance(t, torch.Tensor) and n in ref[j] and not torch.equal(t, ref[j][n]):
            differs.append(n)
    rec = {"step": it, "lane": ln, "input": j, "prev_input_on_lane": prev, "other_lane_input": other, "differs": differs,
           "chunks": []}
    first = None
    for lvl in range(5):
        if final_name(plan, lvl) in differs:
            first = lvl
            break
    line = f"step {it} lane {ln} input {j} prev {prev}: differs {differs}"
    if first is not None:
        name = final_name(plan, first)
        wrong, right = plan.bufs[name], ref[j][name]
        dd = (wrong != right)
        idx = dd.nonzero()
        rows = {}
        for b, c, yy, xx in idx.tolist():
            rows.setdefault((b, c, yy), []).append(xx)
        for (b, c, yy), xl in list(rows.items())[:8]:
            x0, x1 = min(xl), max(xl) + 1
            sl = (b, c, yy, slice(x0, x1))
            wv, rv = wrong[sl].float(), right[sl].float()
            cands = {}
            pre = ref[j].get(f"pre.rgb{first}")
            if pre is not None:
                cands["skip_only(this)"] = pre[sl]
                cands["right-skip(this)"] = rv - pre[sl]
            if prev is not None:
                cands["image(prev frame of lane)"] = ref[prev][name][sl]
                if f"pre.rgb{first}" in ref[prev]:
                    cands["skip(prev frame of lane)"] = ref[prev][f"pre.rgb{first}"][sl]
                    if pre is not None:
                        cands["right-skip(this)+skip(prev)"] = rv - pre[sl] + ref[prev][f"pre.rgb{first}"][sl]
                        cands["right-skip(this)+image(prev)"] = rv - pre[sl] + ref[prev][name][sl]
            if other is not None and name in ref[other]:
                cands["image(other lane's frame)"] = ref[other][name][sl]
            part = ref[j][f"part{first}"][:, b, c, yy, x0:x1]            # (C/8, n)
            ng = part.shape[0]
            for gi in range(ng):
                cands[f"right-part[{gi}]"] = rv - part[gi]
            half = ng // 2
            cands["right-wave0"] = rv - part[:half].sum(0)
            cands["right-wave1"] = rv - part[half:].sum(0)
            cands["zero"] = torch.zeros_like(rv)
            errs = sorted(((float((wv - cv.float()).abs().max()), k) for k, cv in cands.items()))
            scale = float(rv.abs().max())
            rec["chunks"].append({"level": first, "b": b, "plane": c, "y": yy, "x0": x0, "x1": x1,
                                  "wrong": wv.tolist(), "right": rv.tolist(),
                                  "skip": pre[sl].tolist() if pre is not None else None,
                                  "parts": part.tolist(), "best": errs[:4], "scale": scale})
            line += (f"\n    {name} b{b} plane{c} y{yy} (y%8={yy % 8}, y%16={yy % 16}) x{x0}:{x1}  max|right|={scale:.3g} "
                     f"max|wrong-right|={float((wv - rv).abs().max()):.3g}  best: " +
                     ", ".join(f"{k} {e:.2e}" for e, k in errs[:3]))
    print(line, flush=True)
    records.append(rec)


for it in range(steps):
    ln = it % 2
    with (contextlib.nullcontext() if EMU else torch.cuda.stream(streams[ln])):
        if outs[ln] is not None:
            y, j = outs[ln]
            if not EMU:
                streams[ln].synchronize()
            if not torch.equal(y, ref_out[j]):
                bad += 1
                if bad <= MAXDIAG:
                    diagnose(it, ln, j)
        j = it % NIN
        hist[ln].append(j)
        y = eng.forward(xs[j], style, d_s, shared_style=True, use_graph=use_graph and not EMU, lane=ln + 1)
        if EMU and os.environ.get("FLAKE_EMU_CORRUPT") and it == 5:   # self-test of the diagnosis
            pl = lane_plan(ln + 1)
            pl.bufs["rgb3"][0, 0, 3, 16:32] = ref[j]["pre.rgb3"][0, 0, 3, 16:32] if "pre.rgb3" in ref[j] else 0.0
            y = y + 1.0
        outs[ln] = (y.clone(), j)
sync()
print(bb, B, H, W, "graph" if use_graph else "eager", "bad", bad, "of", steps)
if out_json:
    with open(out_json, "w") as f:
        json.dump({"args": sys.argv[1:], "lib": os.environ.get("FLAKE_LIB", "default"), "bad": bad, "steps": steps,
                   "records": records}, f)
