#!/usr/bin/env python
"""Two engine lanes on two streams, alternating batches for many steps, every output compared with the serial result of the
same input (python tools/flake_lanes.py [T|D] B H W STEPS [graph|eager])."""
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
dev = torch.device("cuda:0")
if os.environ.get("FLAKE_LIB"):   # another build of the library (experiments)
    from vtoonify_amd import _lib
    _lib.use_library(os.environ["FLAKE_LIB"])
backbone = "toonify" if bb == "T" else "dualstylegan"
from conftest import load_keys  # noqa: E402
sd = synth.synth_state_dict(load_keys(bb), 0)
eng = VToonifyEngine({k: v.to(dev) for k, v in sd.items()}, backbone, 256, torch.bfloat16, dev)
style = synth.synth_style(seed=5).to(dev)
g = torch.Generator().manual_seed(1)
xs = [torch.randn(B, 22 if bb == "D" else 22, H, W, generator=g).to(dev) for _ in range(6)]
d_s = 0.6 if bb == "D" else None
ref = [eng.forward(x, style, d_s, shared_style=True, use_graph=False).clone() for x in xs]
torch.cuda.synchronize()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
outs = [None, None]
bad = 0
for it in range(steps):
    ln = it % 2
    with torch.cuda.stream(streams[ln]):
        if outs[ln] is not None:
            y, j = outs[ln]
            streams[ln].synchronize()
            if not torch.equal(y, ref[j]):
                bad += 1
                if bad == 1 and os.environ.get("FLAKE_DIAG"):
                    # which plan buffers of this lane differ from a serial rerun of the same input?
                    torch.cuda.synchronize()
                    plan = [p for k, p in eng._plans.items() if k[-1] == ln + 1 and k[0] == B][0]
                    snap = {n: t.clone() for n, t in plan.bufs.items() if isinstance(t, torch.Tensor)}
                    eng.forward(xs[j], style, d_s, shared_style=True, use_graph=use_graph, lane=ln + 1)
                    torch.cuda.synchronize()
                    for n, t in plan.bufs.items():
                        if isinstance(t, torch.Tensor) and n in snap and not torch.equal(t, snap[n]):
                            dd = (t.float() - snap[n].float()).abs()
                            nz = (dd > 0).nonzero()
                            print("   differs:", n, tuple(t.shape), "count", int((dd > 0).sum()), "first idx", nz[0].tolist(), "last idx", nz[-1].tolist())
                if bad <= 5:
                    d = (y.float() - ref[j].float()).abs()
                    print("step", it, "lane", ln, "input", j, "max abs diff", float(d.max()), "count", int((d > 0).sum()))
        j = it % len(xs)
        y = eng.forward(xs[j], style, d_s, shared_style=True, use_graph=use_graph, lane=ln + 1)
        outs[ln] = (y.clone(), j)
torch.cuda.synchronize()
print(bb, B, H, W, "graph" if use_graph else "eager", "bad", bad, "of", steps)
