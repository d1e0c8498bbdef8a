#!/usr/bin/env python
"""First plan buffer of frame 0 that differs between a batch-of-4 call and a single-frame call (VT_BATCH_EXACT=1).  (GPU)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from vtoonify_amd import synth, _lib
from vtoonify_amd.engine import VToonifyEngine
from conftest import load_keys
_lib.use_library(_lib.DEFAULT_LIB)
dev = torch.device("cuda:0")
os.environ["VT_BATCH_EXACT"] = "1"
sd = {k: v.to(dev) for k, v in synth.synth_state_dict(load_keys("D"), 0).items()}
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (144, 200)
x = synth.synth_frames(4, h, w, seed=77).to(dev)
s = synth.synth_style(seed=17).to(dev)
eng = VToonifyEngine(sd, "dualstylegan", 256, torch.bfloat16, dev)
eng.forward(x, s.repeat(4, 1, 1), 0.5, use_graph=False, lane=1)
eng.forward(x[:1].contiguous(), s, 0.5, use_graph=False, lane=2)
torch.cuda.synchronize()
p4 = [p for k, p in eng._plans.items() if k[0] == 4][0]
p1 = [p for k, p in eng._plans.items() if k[0] == 1][0]
for name, b4 in p4.bufs.items():
    b1 = p1.bufs.get(name)
    if b1 is None or b1.shape[1:] != b4.shape[1:] or b4.shape[0] != 4 or b1.shape[0] != 1:
        continue
    eq = torch.equal(b4[:1], b1)
    if not eq:
        d = (b4[:1].float() - b1.float()).abs()
        print(f"{name:16s} {tuple(b4.shape)} differs: max {float(d.max()):.3e} frac {float((d > 0).float().mean()):.3f}")
print("convs of the batch plan:")
for d, info, _, _ in p4.convs:
    print("  ", info.get("kernel"), info.get("sig"), "splitk", info.get("splitk"))
print("convs of the single-frame plan:")
for d, info, _, _ in p1.convs:
    print("  ", info.get("kernel"), info.get("sig"), "splitk", info.get("splitk"))
