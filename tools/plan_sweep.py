#!/usr/bin/env python
"""Per-layer plan search in the regime the benchmark runs in (N frames in flight on N HIP streams).

The built-in heuristics of vt_conv2d (choose_plan) were tuned on single launches; with several frames
in flight what matters is how a launch shares the chip.  This tool measures whole-frame frames/s
(hipGraph replay, `--lanes` frames in flight) while overriding ONE conv geometry at a time through
VToonifyEngine(tile_hints=...), keeps an override when it wins by more than the noise margin, and
writes the resulting table as JSON (usable via VToonifyEngine(tile_hints=) or VT_TILE_HINTS=file).

    python tools/plan_sweep.py [--lanes 3] [--steps 60] [--margin 0.012] [--out gpurun_out/tile_hints.json]
                               [--only SUBSTR] [--passes 1]

Greedy coordinate descent, geometries ordered by their share of the single-stream kernel time.
Outputs stay within the bf16 tolerance of the default plans (only the fp32 summation order changes);
`--check` compares every accepted table against the default output (max-abs relative error printed).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

P = 100000000


def keys(tag):
    with open(os.path.join(REPO, "tests", "golden", f"keys_{tag}.json")) as f:
        return {k: tuple(v) for k, v in json.load(f).items()}


def candidates(sig):
    """tile_hint values worth trying for a conv geometry 'HxW:cin->cout:kKsSdDpP[:nchw]'."""
    geo, chans, kk = sig.split(":")[:3]
    cout = int(chans.split("->")[1])
    k = int(kk[1:kk.index("s")])
    stride = int(kk[kk.index("s") + 1:kk.index("d")])
    dil = int(kk[kk.index("d") + 1:kk.index("p")])
    out = []
    splits = (0, 1, 2, 4, 8)
    if k == 3 and stride == 1:      # patch-resident instances (dil 1 has all tiles, dil 2/4 the 128x128 and 64x64 ones)
        tiles = (256128, 256064, 128128, 128064, 64064, 128016) if dil == 1 else (128128, 64064)
        for t in tiles:
            if t % 1000 > max(cout, 16) * 2 and t % 1000 > 16:
                continue            # tile much wider than the layer
            for s in splits:
                out.append(P + s * 1000000 + t)
    for t in (128128, 128064, 64064, 64128, 128032, 128016):   # 1-D direct-to-LDS / register-staged kernels
        if t % 1000 > max(cout, 16) * 2 and t % 1000 > 16:
            continue
        for s in splits[:4]:
            out.append(2 * P + s * 1000000 + t)
    return out


class Rig:
    def __init__(self, lanes, steps, backbone, H, W, emu=False):
        from vtoonify_amd import _lib, synth
        from vtoonify_amd.engine import VToonifyEngine
        self.emu = emu
        if emu:   # control-flow dry run on the host emulation of the kernels (no timing meaning)
            sys.path.insert(0, os.path.join(REPO, "tests"))
            from emu import build_emu
            _lib.use_library(build_emu.build())
            self.dev = torch.device("cpu")
        else:
            _lib.use_library(_lib.DEFAULT_LIB)
            self.dev = torch.device("cuda:0")
        tag = "D" if backbone == "dualstylegan" else "T"
        sd = synth.synth_state_dict(keys(tag), 0)
        self.eng = VToonifyEngine({k: v.to(self.dev) for k, v in sd.items()}, backbone, 256,
                                  torch.float32 if emu else torch.bfloat16, self.dev)
        self.style = synth.synth_style(seed=17).to(self.dev)
        self.pool = [synth.synth_frames(1, H, W, seed=i).to(self.dev) for i in range(4)]
        self.lanes, self.steps = lanes, steps
        self.streams = None if emu else ([torch.cuda.current_stream(self.dev)] +
                                         [torch.cuda.Stream(self.dev) for _ in range(lanes - 1)])

    def set_hints(self, hints):
        self.eng.tile_hints = dict(hints)
        self.eng._plans.clear()          # plans (buffers, graphs) are rebuilt with the new table
        if not self.emu:
            torch.cuda.empty_cache()

    def step(self, i):
        ln = i % self.lanes
        if self.emu:
            return self.eng.forward(self.pool[i % 4], self.style, 0.5, shared_style=True, lane=ln)
        with torch.cuda.stream(self.streams[ln]):
            return self.eng.forward(self.pool[i % 4], self.style, 0.5, shared_style=True, use_graph=True, lane=ln)

    def fps(self, repeats=3):
        sync = (lambda: None) if self.emu else torch.cuda.synchronize
        for i in range(self.lanes):
            y = self.step(i)
            sync()
        for i in range(0 if self.emu else 2 * self.lanes):
            self.step(i)
        sync()
        best = 0.0
        for _ in range(repeats):
            t0 = time.perf_counter()
            for i in range(self.steps):
                self.step(i)
            sync()
            best = max(best, self.steps / (time.perf_counter() - t0))
        return best, y

    def shares(self):
        """conv geometries of the frame with their share of the single-stream kernel time."""
        self.step(0)
        plan = self.eng.plan_for(1, self.pool[0].shape[2], self.pool[0].shape[3], True, True)
        if self.emu:   # no HIP events on the host: rank by algorithmic flops instead
            per = [(self.eng._info(op[2]), 1e-12 * self.eng._info(op[2]).get("flops", 0)) for op in self.eng.frame_ops(plan)]
        else:
            torch.cuda.synchronize()
            per = self.eng.time_ops(plan, iters=3)
        agg = {}
        for info, ms in per:
            if info.get("name") == "conv":
                a = agg.setdefault(info["sig"], [0.0, 0, info["kernel"]])
                a[0] += ms
                a[1] += 1
        return sorted(agg.items(), key=lambda kv: -kv[1][0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lanes", type=int, default=3)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--margin", type=float, default=0.012, help="relative gain an override must show to be kept")
    ap.add_argument("--out", default=os.path.join(REPO, "gpurun_out", "tile_hints.json"))
    ap.add_argument("--only", default="", help="only geometries containing this substring")
    ap.add_argument("--passes", type=int, default=1)
    ap.add_argument("--top", type=int, default=16, help="geometries to search (by time share)")
    ap.add_argument("--budget-s", type=float, default=600.0)
    ap.add_argument("--backbone", default="dualstylegan")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--emu", action="store_true", help="dry run of the search loop on the CPU emulation (tiny frames)")
    ap.add_argument("--size", type=int, default=256, help="input height = width")
    a = ap.parse_args()
    t_start = time.perf_counter()
    rig = Rig(a.lanes, a.steps, a.backbone, a.size, a.size, emu=a.emu)
    rig.set_hints({})
    base, y0 = rig.fps()
    y0 = y0.clone()
    sigs = rig.shares()
    print(f"default plans: {base:.1f} frames/s with {a.lanes} in flight; {len(sigs)} conv geometries", flush=True)
    for sig, (ms, n, kern) in sigs[:a.top]:
        print(f"   {sig:<36} x{n:2d} {1e3 * ms:8.1f} us  {kern}", flush=True)
    hints, best = {}, base
    log = []
    for ps in range(a.passes):
        for sig, (ms, n, kern) in sigs[:a.top]:
            if a.only and a.only not in sig:
                continue
            for h in candidates(sig):
                if time.perf_counter() - t_start > a.budget_s:
                    break
                trial = dict(hints)
                trial[sig] = h
                try:
                    rig.set_hints(trial)
                    f, y = rig.fps(repeats=2)
                except Exception as e:   # tile not compiled / not eligible for this geometry
                    log.append({"sig": sig, "hint": h, "error": str(e)[:80]})
                    continue
                log.append({"sig": sig, "hint": h, "fps": f})
                if f > best * (1.0 + a.margin):
                    rig.set_hints(trial)
                    f2, y = rig.fps(repeats=3)     # confirm before keeping
                    if f2 > best * (1.0 + a.margin):
                        err = float((y.float() - y0.float()).abs().max() / y0.float().abs().max())
                        print(f"keep {sig} -> {h}: {best:.1f} -> {f2:.1f} frames/s (rel diff vs default output {err:.2e})",
                              flush=True)
                        hints, best = trial, f2
    rig.set_hints(hints)
    final, y = rig.fps()
    rig.set_hints({})
    again, _ = rig.fps()
    res = {"lanes": a.lanes, "default_fps": base, "default_fps_remeasured": again, "tuned_fps": final,
           "hints": hints, "trials": len(log)}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(hints, f, indent=1)
    with open(a.out.replace(".json", "_log.json"), "w") as f:
        json.dump({"summary": res, "log": log}, f)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
