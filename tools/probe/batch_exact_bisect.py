#!/usr/bin/env python
"""Which batch-dependent plan choice breaks 'frame in a batch == frame alone' under VT_BATCH_EXACT=1?  (GPU)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vtoonify_amd import synth, _lib
from vtoonify_amd.engine import VToonifyEngine
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from conftest import load_keys

_lib.use_library(_lib.DEFAULT_LIB)
dev = torch.device("cuda:0")
sd = {k: v.to(dev) for k, v in synth.synth_state_dict(load_keys("D"), 0).items()}
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (144, 200)
x = synth.synth_frames(4, h, w, seed=77).to(dev)
s = synth.synth_style(seed=17).to(dev)
os.environ["VT_BATCH_EXACT"] = "1"
for name, env in (("default", {}), ("P8=0", {"VT_UPBLUR_P8": "0"}), ("TALL=0", {"VT_UPBLUR_TALL": "0"}),
                  ("P8=0 TALL=0", {"VT_UPBLUR_P8": "0", "VT_UPBLUR_TALL": "0"}),
                  ("FULLKW=0", {"VT_FULLKW": "0"}), ("STEM32=0", {"VT_STEM32": "0"}),
                  ("all off", {"VT_UPBLUR_P8": "0", "VT_UPBLUR_TALL": "0", "VT_FULLKW": "0"})):
    for k, v in env.items():
        os.environ[k] = v
    eng = VToonifyEngine(sd, "dualstylegan", 256, torch.bfloat16, dev)
    yb = eng.forward(x, s.repeat(4, 1, 1), 0.5).clone()
    ya = torch.cat([eng.forward(x[i:i + 1].contiguous(), s, 0.5).clone() for i in range(4)])
    d = (yb.float() - ya.float()).abs()
    print(f"{name:14s} equal {bool(torch.equal(yb, ya))} max {float(d.max()):.3e} frames differing {[int(i) for i in range(4) if not torch.equal(yb[i], ya[i])]}")
    for k in env:
        del os.environ[k]
    del eng
