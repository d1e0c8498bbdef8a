#!/usr/bin/env python
"""Face-parsing (BiSeNet) throughput on one MI355X: parsing maps/s for 256x256 frames (the net runs
at 512x512, style_transfer.py:171-172), HBM-resident, hipGraph replay, 1 and 3 frames in flight;
the CPU oracle timed beside it; and the video driver end to end with the maps computed on the GPU
(uint8 frames in host memory -> uint8 1024x1024 frames in host memory, no parsing maps supplied).

usage: python tools/bisenet_bench.py [--steps 100] [--no-cpu] [--no-video]"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def keys(tag):
    with open(os.path.join(REPO, "tests", "golden", f"keys_{tag}.json")) as f:
        return {k: tuple(v) for k, v in json.load(f).items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-video", action="store_true")
    a = ap.parse_args()
    from vtoonify_amd import _lib, synth, video
    from vtoonify_amd.bisenet import BiSeNetEngine
    from vtoonify_amd.engine import VToonifyEngine
    _lib.use_library(_lib.DEFAULT_LIB)
    dev = torch.device("cuda:0")
    bsd = synth.synth_state_dict(keys("bisenet"), 0)
    par = BiSeNetEngine({k: v.to(dev) for k, v in bsd.items()}, 19, torch.bfloat16, dev)
    res = {"what": "BiSeNet parsing maps, 3x256x256 frames -> net at 512x512 -> 19x256x256, bf16"}
    g = torch.Generator().manual_seed(0)
    for B in (1, 4):
        x = (torch.rand(B, 3, 256, 256, generator=g) * 2 - 1).to(dev)
        for lanes in (1, 3):
            streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(lanes - 1)]
            for i in range(2 * lanes):
                with torch.cuda.stream(streams[i % lanes]):
                    par.parsing_maps(x, use_graph=True, lane=i % lanes)
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(a.steps):
                with torch.cuda.stream(streams[i % lanes]):
                    par.parsing_maps(x, use_graph=True, lane=i % lanes)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res[f"maps_per_s_b{B}_lanes{lanes}"] = a.steps * B / dt
            print(f"batch {B} lanes {lanes}: {a.steps * B / dt:8.1f} maps/s ({1e3 * dt / a.steps:.3f} ms/step)", flush=True)
    if not a.no_video:
        sd = synth.synth_state_dict(keys("D"), 0)
        eng = VToonifyEngine({k: v.to(dev) for k, v in sd.items()}, "dualstylegan", 256, torch.bfloat16, dev)
        style = synth.synth_style(seed=17).to(dev)
        rng = np.random.default_rng(0)
        frames = rng.integers(0, 256, (8, 256, 256, 3), dtype=np.uint8)
        for batch, depth in ((1, 3), (4, 2), (4, 3)):
            vt = video.VideoToonifier(eng, style, 0.5, batch_size=batch, depth=depth, parsing_engine=par)
            vt.run(((frames[i % 8], None) for i in range(2 * batch * depth)), lambda i, f: None)
            torch.cuda.synchronize()
            n = 192
            t0 = time.perf_counter()
            vt.run(((frames[i % 8], None) for i in range(n)), lambda i, f: None)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res[f"video_fps_b{batch}_depth{depth}"] = n / dt
            print(f"video (parsing on GPU) batch {batch} depth {depth}: {n / dt:7.1f} frames/s end to end", flush=True)
    if not a.no_cpu:
        from oracle import bisenet_oracle as BO, vtoonify_oracle as O   # the checker, timed as the CPU baseline
        cores = min(os.cpu_count() or 1, 32)
        torch.set_num_threads(cores)
        O.set_backend("torch")
        sdn = synth.to_numpy_sd(bsd)
        xh = (torch.rand(1, 3, 256, 256, generator=g) * 2 - 1).numpy()
        BO.parsing_maps(sdn, xh)
        t0 = time.perf_counter()
        for _ in range(3):
            BO.parsing_maps(sdn, xh)
        dt = (time.perf_counter() - t0) / 3
        res["cpu_baseline"] = {"value": 1.0 / dt, "unit": "maps/s", "cores": cores, "kind": "port",
                               "sample": f"3 frames 3x256x256 through oracle/bisenet_oracle.py (torch-CPU convs), {dt:.3f} s each"}
        print(f"CPU oracle ({cores} threads): {1.0 / dt:.2f} maps/s", flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
