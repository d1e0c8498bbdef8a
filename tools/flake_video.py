#!/usr/bin/env python
"""Repeats the video driver on one engine and counts runs whose frames differ from the frame-by-frame result
(python tools/flake_video.py REPS MODE).  MODE: all | only22 | nograph | sync"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import test_video as TV  # noqa: E402
from vtoonify_amd import video  # noqa: E402

reps = int(sys.argv[1])
mode = sys.argv[2] if len(sys.argv) > 2 else "all"
dev = torch.device("cuda:0")
n, H, W = 11, 64, 96
frames, parsing, eng, style = TV._video_case(dev, n, H, W, 2, 2, torch.bfloat16)
want = TV._expected(eng, style, frames, parsing, dev)
cfgs = ((2, 2),) if mode in ("only22", "overlap") else ((2, 2), (4, 1), (3, 3))
bad = 0
for rep in range(reps):
    for batch, depth in cfgs:
        got = {}
        if mode == "sync":
            torch.cuda.synchronize()
        kw = {"use_graph": False} if mode == "nograph" else {}
        vt = video.VideoToonifier(eng, style, None, batch_size=batch, bgr=True, depth=depth, **kw)
        vt.run(((frames[i], parsing[i]) for i in range(n)), lambda i, fr: got.__setitem__(i, fr.copy()))
        d = [i for i in range(n) if not np.array_equal(got[i], want[i])]
        if d:
            bad += 1
            if bad <= 4:
                print("rep", rep, "batch/depth", batch, depth, "frames differ", d)
                for i in d[:2]:
                    ys, xs, cs = np.nonzero(got[i] != want[i])
                    print("   frame", i, "pixels", len(ys), "rows", ys.min(), ys.max(), "cols", xs.min(), xs.max(), "max delta",
                          int(np.abs(got[i].astype(int) - want[i].astype(int)).max()))
    if mode == "all":
        got = {}
        for r in range(2):
            video.toonify_shard(eng, style, None, lambda i: (frames[i], parsing[i]), n, lambda i, fr: got.__setitem__(i, fr.copy()),
                                batch_size=2, rank=r, world_size=2)
        d = [i for i in range(n) if not np.array_equal(got[i], want[i])]
        if d:
            bad += 1
            print("rep", rep, "shard frames differ", d)
print(mode, "bad", bad, "of", reps)

# ---- do any two live buffers of the engine's plans overlap? ---------------------------------
if mode == "overlap":
    spans = []
    for key, plan in eng._plans.items():
        for name, t in plan.bufs.items():
            if isinstance(t, torch.Tensor) and t.is_cuda and t.numel():
                st = t.untyped_storage()
                spans.append((st.data_ptr(), st.data_ptr() + st.nbytes(), key, name))
    spans.sort()
    nov = 0
    for (a0, a1, k0, n0), (b0, b1, k1, n1) in zip(spans, spans[1:]):
        if b0 < a1 and a0 != b0:
            nov += 1
            print("OVERLAP", k0, n0, hex(a0), hex(a1), "|", k1, n1, hex(b0), hex(b1))
    print("buffers", len(spans), "overlaps", nov)
