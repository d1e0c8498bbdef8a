#!/usr/bin/env python
"""Headline benchmark: frames/s of VToonify-D inference at 1024x1024 output on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (VToonify.forward, model/vtoonify.py:210-277 of the
reference) over one batch of synthetic frames already resident in HBM.  Frames of a video are
independent, so --lanes L (default 3) keeps L steps in flight on L HIP streams, each
with its own plan buffers and hipGraph: step i is issued on stream i % L.  Every step still does
the whole frame; `single_stream` in the JSON line is the same workload with one frame in flight.  The workload at
N=1 is BASELINE.json configs[1]: 22x256x256 frames -> 3x1024x1024, VToonify-D, bf16 compute (fp32 accumulate /
statistics / RGB skip path), seeded synthetic weights; a step is one VToonify.forward over --batch frames (default 4,
the reference's own --batch_size, style_transfer.py:35,176: the video loop calls the model on 4 frames at a time); the
one-frame-per-step rate (round 2's headline) is the `batch1` key, `single_stream` is one step in flight.
Nothing is cached across steps: the style path (T_c/T_s linears, weight modulation +
demodulation, AdaIN gamma/beta) is recomputed every frame like the reference does.

For N>1 every rank processes its own shard of frames (weak scaling: per-GPU work fixed);
the only collective is the one-time RCCL broadcast of weights + style code before the timed
region (vtoonify_amd/frames.py).  Timing: barrier + synchronize on both sides of exactly K
steps, MAX over ranks; value = N*K*batch / that time.

The JSON line also carries
  roofline      the dominant kernel of the frame (largest share of GPU time), its achieved
                ALGORITHMIC flop/s (or byte/s) = work per launch / mean launch duration measured
                with HIP events on the launch stream, against the MI355X peak;
  cpu_baseline  the CPU oracle (oracle/, a restatement of the reference's op_cpu path)
                timed on this box's host cores on a bounded sample (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # dense MFMA bf16, MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3     # fp32-input MFMA = vector rate
PEAK_F32X3_TFLOPS = PEAK_BF16_TFLOPS / 3.0   # f32x3: every fp32 product is three bf16 MFMAs -> a third of the bf16 matrix peak
PEAK_HBM_GBS = 8000.0       # HBM3E spec


def state_shapes(backbone: str):
    """state_dict schema (key -> shape) of VToonify(backbone) without allocating weights."""
    from vtoonify_amd.vtoonify import VToonify
    with torch.device("meta"):
        m = VToonify(backbone=backbone)
    return {k: tuple(v.shape) for k, v in m.state_dict().items()}


def pmc_traffic(kernel_class: str):
    """HBM bytes per launch of a kernel class from the committed PMC pass
    (profiles/*_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of this
    same command, corrected per MI355X_MICROARCH.md; tools/pmc_traffic.py).  None when no pass
    covers the kernel -- counters cannot be collected from inside the timed process."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None, None
    src = "profiles/" + os.path.basename(files[-1]) + " (committed rocprofv3 --pmc pass of this command, not this run)"
    with open(files[-1]) as f:
        table = json.load(f)["kernels"]
    m = re.match(r"(\w+)<(\w+),(\d+)x(\d+)>", kernel_class)
    if not m:
        hit = [v for k, v in table.items() if k.startswith(kernel_class)]
    else:
        name, dt, bm, bn = m.group(1), m.group(2), int(m.group(3)), int(m.group(4))
        tt = "bf16_t" if dt == "bf16" else "float"
        if name == "conv_fullk_kernel":
            lead = f"{name}<{tt}"
        elif name in ("conv_upblur_kernel", "conv_upflat_kernel"):
            lead = f"{name}<{tt}, {bn},"
        elif name == "conv_upblur_rows_kernel":   # (template argument: the input channel count)
            lead = f"{name}<"
        else:
            lead = f"{name}<{tt}, {bm // 16 if name == 'conv_patch_kernel' else bm}, {bn},"
        hit = [v for k, v in table.items() if k.startswith(lead)]
        if name == "conv_patch_kernel":   # ... and the other kernels of the same plan kind: software-pipelined
            # (csrc/conv_patch_pipe.hpp), persistent (conv_patch_persist.hpp), weights resident (conv_patch_resident.hpp)
            for alt in ("conv_patchp_kernel", "conv_patchq_kernel"):
                hit += [v for k, v in table.items() if k.startswith(lead.replace("conv_patch_kernel", alt))]
            hit += [v for k, v in table.items() if k.startswith(f"conv_patchw_kernel<{tt}, {bm // 16}, {bn}>")]
    n = sum(v["launches_sampled"] for v in hit)
    if not n:
        return None, None
    return sum(v["hbm_bytes_per_launch"] * v["launches_sampled"] for v in hit) / n, src



def op_surface(dev, iters=10):
    """Stand-alone rates of the two operators the reference implements as CUDA kernels -- upfirdn2d
    (model/stylegan/op/upfirdn2d_kernel.cu:107-207) and fused_leaky_relu (fused_bias_act_kernel.cu:18-65) --
    through the drop-in surface (vtoonify_amd.op) at the tensor sizes one VToonify-D frame at 22x256x256 gives them
    (SURVEY.md 8a rows a13/a14), bf16 and fp32: ALGORITHMIC bytes (input + output once) / GPU time, `iters` calls per
    hipGraph replay (the same protocol as tools/op_bench.py)."""
    from vtoonify_amd import synth
    from vtoonify_amd.op import fused_leaky_relu, upfirdn2d

    def timeit(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(iters):
                fn()
        g.replay()
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / iters
            best = us if best is None else min(best, us)
        return best

    k = synth.fir_kernel_2d().to(dev)
    rows = []
    for name, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
        esz = 2 if dt == torch.bfloat16 else 4
        for c, s in ((512, 65), (128, 257), (32, 1025)):     # Blur after the transposed conv (model.py:74-90)
            x = torch.randn(1, c, s, s, device=dev).to(dt)
            us = timeit(lambda: upfirdn2d(x, k * 4, pad=(1, 1)))
            rows.append((f"upfirdn2d blur ({c},{s},{s}) {name}", us, (x.numel() + c * (s - 1) ** 2) * esz))
        x = torch.randn(1, 64, 512, 512, device=dev).to(dt)  # Downsample (model.py:53-71, smooth_parsing_map.py:108)
        us = timeit(lambda: upfirdn2d(x, k, down=2, pad=(1, 1)))
        rows.append((f"upfirdn2d down2 (64,512,512) {name}", us, (x.numel() + 64 * 256 * 256) * esz))
        x = torch.randn(1, 32, 512, 512, device=dev).to(dt)  # Upsample (model.py:32-50) on a feature-sized tensor
        us = timeit(lambda: upfirdn2d(x, k * 4, up=2, pad=(2, 1)))
        rows.append((f"upfirdn2d up2 (32,512,512) {name}", us, (x.numel() + 32 * 1024 * 1024) * esz))
        for c, s in ((512, 32), (128, 256), (32, 1024)):     # FusedLeakyReLU (op/fused_act.py:104-119)
            x = torch.randn(1, c, s, s, device=dev).to(dt)
            b = torch.randn(c, device=dev).to(dt)
            us = timeit(lambda: fused_leaky_relu(x, b))
            rows.append((f"fused_leaky_relu ({c},{s},{s}) {name}", us, (2 * x.numel() + c) * esz))
    for s in (32, 512):                                       # Upsample of the fp32 RGB skip planes
        x = torch.randn(1, 3, s, s, device=dev)
        us = timeit(lambda: upfirdn2d(x, k * 4, up=2, pad=(2, 1)))
        rows.append((f"upfirdn2d up2 (3,{s},{s}) fp32", us, (x.numel() + 12 * s * s) * 4))
    return {"unit": "GB/s", "peak": PEAK_HBM_GBS, "protocol": f"{iters} calls per hipGraph replay, best of 3 replays",
            "rows": [{"op": n, "us": round(us, 2), "gbs": round(nb / us / 1e3, 1),
                      "frac": round(nb / us / 1e3 / PEAK_HBM_GBS, 4)} for n, us, nb in rows]}


def end_to_end(backbone, H, W, frames_per_step, step_s, dtype):
    dual = backbone == "dualstylegan"
    px = H * W / 65536.0
    gflop = (459.1 if dual else 400.5) * px
    act_m, w_m = ((1005.5, 98.2) if dual else (876.3, 59.4))
    esz = 2 if dtype == "bf16" else 4
    gbytes = (act_m * px + w_m) * esz / 1e3
    tf = gflop * frames_per_step / step_s / 1e3
    peak = PEAK_BF16_TFLOPS if dtype == "bf16" else (PEAK_F32X3_TFLOPS if dtype == "fp32x3" else PEAK_F32_TFLOPS)
    return {"gflop_per_frame": round(gflop, 1), "gbyte_per_frame": round(gbytes, 3), "tflops": round(tf, 1),
            "frac_of_mfma_peak": round(tf / peak, 4), "hbm_gbs": round(gbytes * frames_per_step / step_s, 1),
            "frac_of_hbm_peak": round(gbytes * frames_per_step / step_s / PEAK_HBM_GBS, 4),
            "what": "SURVEY.md 8(d) algorithmic FLOPs and op-granularity bytes of the whole step / wall time of the step"}


def kernel_table(eng, plan, dtype, iters, emu=False, want_ops=False, peak_tf=None):
    """Per-kernel-class table of one frame (HIP events on the launch stream around every launch, engine.time_ops)
    and the `roofline` object of the class with the largest share of GPU time.  `peak_tf` overrides the matrix peak the
    fractions are taken against (f32x3 runs on the bf16 pipes at three MFMAs per product: 2500 / 3 TFLOP/s, not the fp32 peak)."""
    if emu:   # no HIP events on the host: one conv entry stands in for the table
        per_op = [(next(info for _, info, _, _ in plan.convs), 1.0)]
    else:
        per_op = eng.time_ops(plan, iters=iters, with_style=True)
    classes = {}
    for info, ms in per_op:
        c = classes.setdefault(info["kernel"], {"ms": 0.0, "launches": 0, "flops": 0, "bytes": 0})
        c["ms"] += ms
        c["launches"] += 1
        c["flops"] += info["flops"]
        c["bytes"] += info["bytes"]
    frame_ms = sum(c["ms"] for c in classes.values())
    if peak_tf is None:
        peak_tf = PEAK_BF16_TFLOPS if dtype == torch.bfloat16 else PEAK_F32_TFLOPS
    rows = []
    for name, c in sorted(classes.items(), key=lambda kv: -kv[1]["ms"]):
        sec = c["ms"] * 1e-3
        tf = c["flops"] / sec / 1e12 if sec > 0 else 0.0
        gbs = c["bytes"] / sec / 1e9 if sec > 0 else 0.0
        # which roof bounds this kernel: compare its arithmetic intensity with the ridge
        ai = c["flops"] / max(c["bytes"], 1)
        bound = "mfma" if ai > peak_tf * 1e12 / (PEAK_HBM_GBS * 1e9) else "hbm"
        rows.append({"kernel": name, "launches": c["launches"], "ms_per_step": c["ms"],
                     "share": c["ms"] / frame_ms if frame_ms else 0.0,
                     "avg_launch_us": 1e3 * c["ms"] / c["launches"], "bound": bound,
                     "tflops": tf, "gbs": gbs,
                     "frac": (tf / peak_tf) if bound == "mfma" else (gbs / PEAK_HBM_GBS)})
    dom = rows[0]
    if dom["bound"] == "mfma":
        roofline = {"bound": "mfma", "achieved": dom["tflops"], "peak": peak_tf, "unit": "TFLOP/s"}
    else:
        roofline = {"bound": "hbm", "achieved": dom["gbs"], "peak": PEAK_HBM_GBS, "unit": "GB/s"}
    roofline["frac"] = roofline["achieved"] / roofline["peak"]
    tr, src = pmc_traffic(dom["kernel"])
    roofline["traffic"] = tr
    roofline["traffic_source"] = src
    # durations: hipEvent pairs around every launch minus HALF the median cost of an empty event pair on this
    # box (`event_gap_us`: two event packets, one of which overlaps a kernel) -- the correction that makes
    # these averages agree with rocprofv3's of the same launches; raw sum kept as `kernel_sum_ms_raw`
    roofline["timing"] = "hipEvent pair per launch minus half the empty-pair gap"
    roofline["event_gap_us"] = 1e3 * getattr(eng, "event_gap_ms", 0.0)
    roofline["kernel_sum_ms_raw"] = sum(getattr(eng, "last_raw_ms", []))
    roofline.update({"kernel": dom["kernel"], "launches_per_step": dom["launches"],
                     "avg_launch_us": dom["avg_launch_us"], "share_of_step": dom["share"],
                     "kernel_sum_ms_per_step": frame_ms})
    return (rows, roofline, per_op) if want_ops else (rows, roofline)


def video_rate(eng, style, d_s, H, W, batch, use_graph, n_frames=96):
    """End-to-end frames/s of vtoonify_amd.video.VideoToonifier: host uint8 frames (+ fp32 parsing
    maps) -> H2D -> pack -> forward -> unpack -> D2H -> sink, double-buffered.  Synthetic frames."""
    import numpy as np
    from vtoonify_amd import video
    g = np.random.default_rng(0)
    frames = g.integers(0, 256, (8, H, W, 3), dtype=np.uint8)
    parsing = (g.standard_normal((8, 19, H, W)) * 4).astype(np.float32)
    vt = video.VideoToonifier(eng, style, d_s, batch_size=batch, bgr=True, depth=3, use_graph=use_graph)
    sink = lambda i, fr: None
    vt.run(((frames[i % 8], parsing[i % 8]) for i in range(2 * batch)), sink)   # warm-up (plans, graph, pinned buffers)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    vt.run(((frames[i % 8], parsing[i % 8]) for i in range(n_frames)), sink)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": n_frames / dt, "unit": "frames/s", "frames": n_frames, "batch": batch,
            "what": "host uint8 BGR frames + fp32 parsing maps in, uint8 BGR frames out (pinned, 3 batches in flight)"}


def pipeline_rate(eng, style, d_s, H, W, batch, use_graph, dev, n_frames=96):
    """The decode-free per-frame pipeline of the reference's video loop (style_transfer.py:166-177): host uint8 frames in ->
    BiSeNet face parsing at twice the frame size -> x_p / 16 concatenated with the normalised frame -> VToonify -> clamp ->
    uint8 frames out, through vtoonify_amd.video.VideoToonifier with the parsing net on the GPU (vtoonify_amd.bisenet).
    Synthetic frames, seeded synthetic BiSeNet weights (missing 5)."""
    import numpy as np
    from vtoonify_amd import synth, video
    from vtoonify_amd.bisenet import BiSeNetEngine
    with open(os.path.join(REPO, "tests", "golden", "keys_bisenet.json")) as f:
        bshapes = {k: tuple(v) for k, v in json.load(f).items()}
    par = BiSeNetEngine({k: v.to(dev) for k, v in synth.synth_state_dict(bshapes, 0).items()}, 19, torch.bfloat16, dev)
    g = np.random.default_rng(0)
    frames = g.integers(0, 256, (8, H, W, 3), dtype=np.uint8)
    vt = video.VideoToonifier(eng, style, d_s, batch_size=batch, bgr=True, depth=3, use_graph=use_graph, parsing_engine=par)
    sink = lambda i, fr: None
    vt.run(((frames[i % 8], None) for i in range(2 * batch)), sink)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    vt.run(((frames[i % 8], None) for i in range(n_frames)), sink)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": n_frames / dt, "unit": "frames/s", "frames": n_frames, "batch": batch,
            "what": f"host uint8 BGR frames in -> BiSeNet (bf16) at {2 * H}x{2 * W} -> VToonify-D -> uint8 BGR {4 * H}x{4 * W} frames "
                    f"out (pinned buffers, 3 batches in flight); no video decode / encode"}


def cpu_reference(backbone: str, height: int, width: int, budget_s: float, cores: int):
    """The reference's OWN op_cpu path (model/vtoonify.py:210-277 over model/stylegan/op_cpu) timed in a subprocess -- only
    where the reference is mounted (VTOONIFY_REFERENCE or /root/reference: the authoring container; the GPU boxes have
    neither, and nothing else in this file reads it).  None when it is absent or fails."""
    import subprocess
    ref = os.environ.get("VTOONIFY_REFERENCE", "/root/reference")
    if not os.path.isfile(os.path.join(ref, "model", "vtoonify.py")):
        return None
    try:
        r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "cpu_reference.py"), "--threads", str(cores),
                            "--height", str(height), "--width", str(width), "--backbone", backbone, "--budget", str(budget_s)],
                           capture_output=True, text=True, timeout=20 * budget_s + 300)
        d = json.loads([ln for ln in r.stdout.split("\n") if ln.startswith("{")][-1])
    except Exception:
        return None
    reps, h, w = d["reps_s"], d["h"], d["w"]
    dt = sorted(reps)[len(reps) // 2]
    return {"value": 1.0 / (dt * (height * width) / (h * w)), "unit": "frames/s", "cores": cores, "kind": "reference",
            "repetitions_s": [round(t, 3) for t in reps],
            "sample": f"1 frame 22x{h}x{w} -> 3x{4 * h}x{4 * w} fp32 through {ref}/model/vtoonify.py over model/stylegan/op_cpu "
                      f"(tools/cpu_reference.py, own process), median of {len(reps)} runs = {dt:.2f} s, scaled "
                      f"x{(height * width) // (h * w)} in pixels to the 22x{height}x{width} workload"}


def cpu_baseline(backbone: str, height: int, width: int, budget_s: float):
    """Time the CPU oracle on the host cores.  Sample: ONE frame of the benchmark workload when
    that fits the budget, otherwise a centre crop scaled to it (cost is linear in H*W).  Where the reference itself is
    mounted, its own op_cpu path is timed instead (kind "reference") and the oracle's time is kept beside it."""
    import numpy as np
    from oracle import vtoonify_oracle as O  # the checker, timed as the CPU baseline
    from vtoonify_amd import synth
    # threads actually used: oneDNN convolutions scale to a few tens of cores on these shapes and
    # fall off a cliff when a 256-thread pool is woken for every small op
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    O.set_backend("torch")  # dense contractions through F.conv2d on CPU, like the reference's op_cpu path
    sd = synth.to_numpy_sd(synth.synth_state_dict(state_shapes(backbone), 0))
    s = synth.synth_style(seed=17).numpy()
    # size the sample from a 64x64 probe (cost is ~linear in H*W): the full frame if it fits
    xp = synth.synth_frames(1, 64, 64, seed=1).numpy()
    O.vtoonify_forward(sd, xp, s, 0.5, backbone)  # warm-up (oneDNN primitive caches)
    t0 = time.perf_counter()
    O.vtoonify_forward(sd, xp, s, 0.5, backbone)
    t_probe = time.perf_counter() - t0
    h, w = height, width
    while t_probe * (h * w) / (64 * 64) > budget_s and h * w > 64 * 64:
        h, w = max(h // 2, 64), max(w // 2, 64)
    reps = []
    if (h, w) == (64, 64):
        reps = [t_probe]
    else:
        # median of up to three repetitions inside the budget (one sample moved by +-20 % between runs)
        x = synth.synth_frames(1, h, w, seed=2).numpy()
        t_all = time.perf_counter()
        while len(reps) < 3 and (not reps or time.perf_counter() - t_all + reps[-1] < budget_s):
            t0 = time.perf_counter()
            y = O.vtoonify_forward(sd, x, s, 0.5, backbone)
            reps.append(time.perf_counter() - t0)
        assert np.isfinite(y).all()
    dt = sorted(reps)[len(reps) // 2]
    # frames/s of the benchmark workload: scale the sample linearly in pixels
    fps = 1.0 / (dt * (height * width) / (h * w))
    real = cpu_reference(backbone, height, width, budget_s, cores)
    if real is not None:
        real["oracle_port"] = {"value": fps, "repetitions_s": [round(t, 3) for t in reps]}
        return real
    return {"value": fps, "unit": "frames/s", "cores": cores, "kind": "port",
            "backend": getattr(O, "BACKEND", "numpy"),
            "restates": "model/stylegan/op_cpu/upfirdn2d.py:20-60, op_cpu/fused_act.py:23-34, model/stylegan/model.py:259-306 "
                        "(ModulatedConv2d, fused branch), :32-90 (Blur / Upsample), model/dualstylegan.py:6-45, "
                        "model/vtoonify.py:92-128,210-277 -- the reference tree is not mounted on this box, so its own op_cpu "
                        "path cannot be timed here (kind stays \"port\")",
            "cpu_branch": cpu_branch_rate(backbone, h, w, height, width, cores),
            "repetitions_s": [round(t, 3) for t in reps],
            "port_vs_reference": "profiles/r04_cpu_port_vs_reference.txt (authoring container, 8 idle cores, alternating): the "
                                 "reference's own op_cpu path (model/vtoonify.py:210-277 over model/stylegan/op_cpu) 1.72 s per frame, this "
                                 "oracle 2.60 s -- the port UNDERSTATES the reference's CPU rate by about 1.5x",
            "sample": f"1 frame 22x{h}x{w} -> 3x{4 * h}x{4 * w} fp32 through oracle/vtoonify_oracle.py, median of "
                      f"{len(reps)} runs = {dt:.2f} s, scaled x{(height * width) // (h * w)} in pixels to the "
                      f"22x{height}x{width} workload"}


def cpu_branch_rate(backbone, h, w, height, width, cores):
    """The drop-in's own CPU-tensor path (`style_transfer.py --cpu`: vtoonify_amd/eager.py over op/native.py -- the reference's
    eager operator sequence on F.conv2d / F.conv_transpose2d, i.e. the same torch calls its op_cpu path makes) on the same
    sample and cores, reported BESIDE the oracle's time, never as the baseline value."""
    try:
        from vtoonify_amd import _lib, synth
        from vtoonify_amd.eager import EagerVToonify
        if _lib.emulation_injected():
            return None
        sd = synth.synth_state_dict(state_shapes(backbone), 0)
        net = EagerVToonify(sd, backbone, 256)
        x, s = synth.synth_frames(1, h, w, seed=2), synth.synth_style(seed=17)
        with torch.no_grad():
            net.forward(x[:, :, :64, :64].contiguous(), s, 0.5)   # warm-up
            t0 = time.perf_counter()
            net.forward(x, s, 0.5)
            dt = time.perf_counter() - t0
        return {"value": 1.0 / (dt * (height * width) / (h * w)), "unit": "frames/s", "cores": cores, "seconds": round(dt, 3),
                "what": f"vtoonify_amd.eager.EagerVToonify on CPU tensors (the module's --cpu path), 1 frame 22x{h}x{w}, one run"}
    except Exception as e:   # an extra: never lose the line over it
        return {"error": f"{type(e).__name__}: {e}"}


def pin_to_gpu_numa_node(local_rank: int, ws: int):
    """One process per GPU: keep this rank's host threads on the NUMA node its GPU hangs off and cap the
    intra-op pool (8 ranks x a 256-thread default pool oversubscribe the host).  Best effort; returns what
    was done for the JSON line."""
    info = {"numa_node": None, "host_threads": None}
    if ws == 1:   # a single rank owns the host (and the cpu_baseline leg wants all of its cores)
        return info
    try:
        pr = torch.cuda.get_device_properties(local_rank)
        bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
            node = int(f.read().strip())
        if node >= 0:
            with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
                cpus = set()
                for part in f.read().strip().split(","):
                    a, _, b = part.partition("-")
                    cpus.update(range(int(a), int(b or a) + 1))
            if cpus:
                os.sched_setaffinity(0, cpus)
                info["numa_node"] = node
    except Exception:
        pass
    try:
        n = max(1, min(16, len(os.sched_getaffinity(0)) // max(1, ws if info["numa_node"] is None else 1)))
        torch.set_num_threads(n)
        info["host_threads"] = n
    except Exception:
        pass
    return info


def timed_blocks(step, steps, dev, min_seconds=1.0, max_blocks=64):
    """Time blocks of EXACTLY `steps` steps, each bracketed by barrier + synchronize on both sides, MAX over
    ranks per block; blocks are repeated until >= min_seconds of timed work (same count on every rank: it is
    derived from the reduced time of the first block).  Returns (median block seconds, [block seconds])."""
    def one():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        el = time.perf_counter() - t0
        if dist.is_initialized():
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el
    times = [one()]
    n = int(min(max_blocks, max(1, -(-min_seconds // max(times[0], 1e-6)))))
    for _ in range(n - 1):
        times.append(one())
    srt = sorted(times)
    return srt[len(srt) // 2], times


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)    # ~0.3 s of GPU time at 1.4 ms per step
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4,
                    help="frames per step per GPU (default 4: the reference's own --batch_size, style_transfer.py:35,176; "
                         "the one-frame-per-step rate is reported as `batch1`)")
    ap.add_argument("--height", type=int, default=256, help="input height (output is 4x)")
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--backbone", default="dualstylegan", choices=["dualstylegan", "toonify"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--d-s", type=float, default=0.5)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--lanes", type=int, default=int(os.environ.get("VT_BENCH_LANES", "3")),
                    help="frames in flight per GPU: step i runs on HIP stream i %% lanes with its own plan buffers")
    ap.add_argument("--tile-hints", default=os.environ.get("VT_TILE_HINTS", ""),
                    help="JSON table {conv geometry: tile_hint} from tools/plan_sweep.py (default: built-in heuristics)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-video", action="store_true", help="skip the PCIe-inclusive video-driver measurement")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra keys (batch4, module_call, config3); `value` is unaffected")
    ap.add_argument("--dry-run-emu", action="store_true",
                    help="TEST MODE (never a measurement): the host-emulation build of the kernels on CPU tensors with the "
                         "gloo backend, so the multi-rank control flow of this file (collectives, equal block counts, "
                         "extra keys) can run under torch.distributed.run without GPUs.  The JSON line is marked "
                         '"data": "dry-run"')
    ap.add_argument("--min-seconds", type=float, default=1.0,
                    help="repeat the --steps block until this much timed work has run; value = median block")
    ap.add_argument("--cpu-budget", type=float, default=25.0, help="seconds of CPU work for cpu_baseline")
    ap.add_argument("--op-iters", type=int, default=5, help="instrumented frames for per-kernel timing")
    ap.add_argument("--kernels", action="store_true", help="also print the per-kernel table (stderr)")
    args = ap.parse_args()

    from vtoonify_amd import _lib, frames, synth
    from vtoonify_amd.engine import VToonifyEngine

    emu = args.dry_run_emu
    rank, local_rank, ws = frames.init("gloo" if emu else None)
    if ws != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={ws}: launch with torch.distributed.run")
    if emu:
        sys.path.insert(0, os.path.join(REPO, "tests"))
        from emu import build_emu
        _lib.use_library(build_emu.build())
        dev = torch.device("cpu")
        host = {"numa_node": None, "host_threads": None}
        torch.cuda.synchronize = lambda *a, **k: None          # no device work to wait for
        torch.cuda.current_stream = lambda *a, **k: None
        import contextlib
        torch.cuda.stream = lambda s: contextlib.nullcontext()
        torch.cuda.Stream = lambda *a, **k: None
        args.no_video, args.no_cpu_baseline, args.no_graph = True, True, True
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the package has no CPU path)")
        _lib.use_library(_lib.DEFAULT_LIB)
        assert not _lib.is_emulation()
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        host = pin_to_gpu_numa_node(local_rank, ws)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    B, H, W = args.batch, args.height, args.width

    # ---- weights + style: synthesised on rank 0, one RCCL broadcast (frames.py) ------------
    shapes = state_shapes(args.backbone)
    sd = synth.synth_state_dict(shapes, 0) if rank == 0 else None
    t0 = time.perf_counter()
    sd_dev = frames.broadcast_state_dict(shapes, sd, dev, skip_unused=True)
    style, d_s = frames.broadcast_style(synth.synth_style(seed=17) if rank == 0 else None,
                                        args.d_s if rank == 0 else None, dev)
    torch.cuda.synchronize()
    t_bcast = time.perf_counter() - t0
    del sd
    hints = None
    if args.tile_hints:
        with open(args.tile_hints) as f:
            hints = {k: int(v) for k, v in json.load(f).items()}
    eng = VToonifyEngine(sd_dev, args.backbone, 256, dtype, dev, tile_hints=hints)
    use_graph = not args.no_graph

    # ---- this rank's shard of the synthetic video, resident in HBM -------------------------
    pool = [synth.synth_frames(B, H, W, seed=1000 * rank + i).to(dev) for i in range(4)]

    lanes = max(1, args.lanes)
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(lanes - 1)]

    def step(i):
        ln = i % lanes
        with torch.cuda.stream(streams[ln]):
            return eng.forward(pool[i % len(pool)], style, d_s, shared_style=True, use_graph=use_graph, lane=ln, borrow=True)

    for i in range(max(args.warmup, lanes)):
        y = step(i)
        if i < lanes:
            torch.cuda.synchronize()   # plan construction + graph capture of each lane, one at a time
    torch.cuda.synchronize()
    assert tuple(y.shape) == (B, 3, 4 * H, 4 * W) and bool(torch.isfinite(y).all())
    # fingerprint of the last warm-up frame: lets two runs (A/B switches, boxes) be compared for equal results
    yf = y.float()
    checksum = {"mean_abs": float(yf.abs().mean()), "max_abs": float(yf.abs().max()),
                "samples": [float(v) for v in yf.flatten()[:: max(1, yf.numel() // 8)][:8]]}

    elapsed, blocks = timed_blocks(step, args.steps, dev, args.min_seconds)

    # the headline's lane-0 plan, held from here on: the extras below build more plans than the engine's LRU keeps
    # (ADVICE r3), and the per-kernel pass at the end must time THIS plan, not a rebuilt one
    headline_plan = eng.plan_for(B, H, W, True, d_s != 0.0) if rank == 0 else None

    # the same workload with ONE frame in flight (latency view; not `value`) -- measured before the extras, while the
    # headline's plans and graphs are the ones in the cache (no rebuild / capture inside a timed repetition)
    single = None
    module_call = None
    reps = 1 if emu else 5
    if rank == 0 and (lanes > 1 or emu):
        n1 = min(args.steps, 1 if emu else 50)
        def s1(i):
            return eng.forward(pool[i % len(pool)], style, d_s, shared_style=True, use_graph=use_graph, lane=0, borrow=True)
        s1(0)   # warm (a no-op when lane 0's plan is resident)
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t1 = time.perf_counter()
            for i in range(n1):
                s1(i)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t1)
        t1 = sorted(ts)[len(ts) // 2]
        single = {"value": n1 * B / t1, "unit": "frames/s", "steps": n1, "ms_per_step": 1e3 * t1 / n1,
                  "steps_in_flight": 1, "frames_per_step": B, "blocks": len(ts)}

    def lanes_rate(bb, hh, ww, nsteps, engine=None, ds=None, n_lanes=None, note=None):
        """frames/s of another (batch, size, engine, style degree) on this rank's GPU, `lanes` steps in flight, all ranks."""
        e = engine or eng
        dd = d_s if ds is None else ds
        nl = n_lanes or lanes
        while len(streams) < nl and not emu:   # `--lanes 1` with an extra that wants two steps in flight
            streams.append(torch.cuda.Stream(dev))
        pl = [synth.synth_frames(bb, hh, ww, seed=5000 + 1000 * rank + i).to(dev) for i in range(2)]
        def st(i):
            ln = i % nl
            with torch.cuda.stream(streams[ln] if ln < len(streams) else None):
                return e.forward(pl[i % 2], style, dd, shared_style=True, use_graph=use_graph, lane=ln, borrow=True)
        for i in range(nl if emu else nl + 2):
            st(i)
            if i < nl:
                torch.cuda.synchronize()
        el, bl = timed_blocks(st, nsteps, dev, min(args.min_seconds, 0.5))
        r = {"value": ws * nsteps * bb / el, "unit": "frames/s", "frames_per_step_per_gpu": bb,
             "ms_per_step": 1e3 * el / nsteps, "steps": nsteps, "blocks": len(bl), "frames_in_flight_per_gpu": nl,
             "workload": f"22x{hh}x{ww} -> 3x{4 * hh}x{4 * ww}"}
        if note:
            r["what"] = note
        return r

    extras = {}
    if not args.no_extras:
        # the reference's default --batch_size 4 (style_transfer.py:35) on the headline frame size, and BASELINE
        # config 3's per-rank step (4 frames of 22x144x256; 960 frames over the job = 240 / world_size steps)
        if emu:   # same control flow, toy sizes
            extras["batch1"] = lanes_rate(1, H, W, 2)
            extras["config3"] = lanes_rate(2, 16, 24, 2)
            extras["config5"] = {"D_16x24": lanes_rate(1, 16, 24, 2)}
        else:
            if B != 4:
                extras["batch4"] = lanes_rate(4, H, W, 24)
            if B != 1:   # one frame per step (round 2's headline workload), same frames in flight
                extras["batch1"] = lanes_rate(1, H, W, 60)
            if B < 16:  # what a caller gets from a larger --batch_size (style_transfer.py:35): 16 frames per step, 2 steps in flight
                extras["batch16"] = lanes_rate(16, H, W, 8, n_lanes=2)
            extras["config3"] = lanes_rate(4, 144, 256, max(8, 240 // ws))
            # BASELINE config 5: 1536x1536 output and the demo's nominal non-square, non-power-of-two crop
            # (vtoonify_model.py:250), VToonify-D, one frame per step
            extras["config5"] = {"D_1536x1536": lanes_rate(1, 384, 384, 24),
                                 "D_1440x1600": lanes_rate(1, 360, 400, 24)}
    if not args.no_extras and ws == 1 and not emu:
        # BASELINE config 4: VToonify-T (Toonify backbone) at 1024x1024 and the style-degree sweep.  T ignores d_s
        # (model/vtoonify.py:238,257); on D d_s = 0 skips the 12 AdaResBlock convs (model/dualstylegan.py:40-41)
        shapes_t = state_shapes("toonify")
        eng_t = VToonifyEngine({k: v.to(dev) for k, v in synth.synth_state_dict(shapes_t, 0).items()}, "toonify", 256,
                               dtype, dev)
        c4 = {"T_1024x1024": lanes_rate(1, 256, 256, 40, engine=eng_t, note="VToonify-T, d_s ignored by the backbone")}
        del eng_t
        c4["D_d_s_sweep"] = {f"{v:g}": round(lanes_rate(1, 256, 256, 40, ds=v)["value"], 1) for v in (0.0, 0.25, 0.5, 0.75, 1.0)}
        extras["config4"] = c4
        # throughput in the reference's own precision: the fp32 engine (exact-fp32 MFMA, the parity mode)
        if dtype != torch.float32:
            eng32 = VToonifyEngine(sd_dev, args.backbone, 256, torch.float32, dev)
            f32 = lanes_rate(1, H, W, 16, engine=eng32, note="fp32 end to end (v_mfma_f32_16x16x4_f32), the reference's precision")
            f32["single_stream"] = lanes_rate(1, H, W, 12, engine=eng32, n_lanes=1)["value"]
            if B > 1:   # ... and at the headline's frames per step (missing 3)
                f32["headline_batch"] = lanes_rate(B, H, W, 8, engine=eng32,
                                                   note=f"fp32 end to end, {B} frames per step like `value`")
            rows32, roof32 = kernel_table(eng32, eng32.plan_for(1, H, W, True, d_s != 0.0), torch.float32, 2)
            f32["roofline"] = roof32
            f32["kernels"] = [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()} for r in rows32[:4]]
            extras["fp32"] = f32
            del eng32
            # ... and the same fp32 tensors with every conv product as three bf16 MFMAs (VToonifyEngine(x3=True), DESIGN.md
            # 4.1i): the reference's precision (4e-5 of max|y| against the fp32 oracle, bar 1e-4) on the bf16 matrix cores
            eng3 = VToonifyEngine(sd_dev, args.backbone, 256, torch.float32, dev, x3=True)
            f3 = lanes_rate(1, H, W, 16, engine=eng3, note="fp32 tensors, convolutions as 3 bf16 MFMAs per product "
                                                             "(operands split into bf16 head + remainder in registers)")
            f3["single_stream"] = lanes_rate(1, H, W, 12, engine=eng3, n_lanes=1)["value"]
            if B > 1:
                f3["headline_batch"] = lanes_rate(B, H, W, 8, engine=eng3, note=f"{B} frames per step like `value`")
            rows3, roof3 = kernel_table(eng3, eng3.plan_for(1, H, W, True, d_s != 0.0), torch.float32, 2,
                                        peak_tf=PEAK_F32X3_TFLOPS)
            f3["peak"] = {"tflops": round(PEAK_F32X3_TFLOPS, 1), "why": "three bf16 MFMAs per fp32 product: a third of the dense bf16 "
                          "matrix peak (the kernels that have no x3 instance run exact fp32 and are priced the same way here)"}
            f3["kernels"] = [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()} for r in rows3[:6]]
            extras["fp32x3"] = f3
            del eng3
        # the headline workload with plans from the per-image geometry only (VT_BATCH_EXACT=1: a frame inside a batch equals
        # the frame alone bit for bit): fresh engine, the switch is read when its plans are built
        if B > 1 and os.environ.get("VT_BATCH_EXACT") != "1":
            os.environ["VT_BATCH_EXACT"] = "1"
            try:
                eng_x = VToonifyEngine(sd_dev, args.backbone, 256, dtype, dev)
                extras["batch_exact"] = lanes_rate(B, H, W, 24, engine=eng_x,
                                                   note="VT_BATCH_EXACT=1: no plan choice that depends on the batch's rounding")
                del eng_x
            finally:
                del os.environ["VT_BATCH_EXACT"]
        torch.cuda.empty_cache()
        extras["op_surface"] = op_surface(dev)

    if rank == 0 and not args.no_extras:
        # through the drop-in module: VToonify(...).load_state_dict(...); model(x, s_w.repeat(B,1,1), d_s=...)
        # exactly as style_transfer.py:62-64,176 calls it (hipGraph replay by default, one frame in flight)
        from vtoonify_amd.vtoonify import VToonify
        m = VToonify(backbone=args.backbone, compute_dtype=dtype)
        m.load_state_dict({k: v for k, v in sd_dev.items()})
        m = m.to(dev)
        sw = style.repeat(B, 1, 1)
        nwarm = 1 if emu else 5
        for i in range(nwarm):
            ym = m(pool[i % len(pool)], sw, d_s=d_s)
        torch.cuda.synchronize()
        assert torch.equal(ym, eng.forward(pool[(nwarm - 1) % len(pool)], style, d_s, shared_style=True,
                                           use_graph=use_graph, lane=0))
        n1 = min(args.steps, 1 if emu else 50)

        def time_module():
            ts = []
            for _ in range(reps):
                t1 = time.perf_counter()
                for i in range(n1):   # a NEW style tensor per call, as style_transfer.py:176 writes it
                    m(pool[i % len(pool)], style.repeat(B, 1, 1), d_s=d_s)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t1)
            return sorted(ts)[len(ts) // 2]
        t1 = time_module()
        module_call = {"value": n1 * B / t1, "unit": "frames/s", "steps": n1, "ms_per_step": 1e3 * t1 / n1, "precision": m.precision,
                       "what": f"VToonify.__call__(x, s_w.repeat(B,1,1), d_s=...) of the drop-in module, B = {B}, one call in "
                               f"flight, output copied out of the plan like a fresh tensor; the module's style gate is on "
                               f"(the style path is skipped on the device while the style rows and d_s do not change -- "
                               f"bit-identical outputs)"}
        # the same module with the gate off: the style path recomputed on every call, like `value`
        os.environ["VT_STYLE_GATE"] = "0"
        m.invalidate()
        for i in range(nwarm):
            ym2 = m(pool[i % len(pool)], sw, d_s=d_s)
        torch.cuda.synchronize()
        assert torch.equal(ym2, ym)
        t2 = time_module()
        os.environ.pop("VT_STYLE_GATE", None)
        module_call["recompute"] = {"value": n1 * B / t2, "ms_per_step": 1e3 * t2 / n1,
                                    "what": "VT_STYLE_GATE=0: style path recomputed on every call"}
        del m

    result = None
    if rank == 0:
        fps = ws * args.steps * B / elapsed
        # ---- per-kernel timing (HIP events on the launch stream), dominant kernel ----------
        plan = headline_plan
        rows, roofline, per_op = kernel_table(eng, plan, dtype, max(1, args.op_iters), emu=emu, want_ops=True)
        if args.kernels:
            for r in rows:
                print(f"{r['kernel']:<36} n={r['launches']:3d} {r['ms_per_step']:8.3f} ms {100 * r['share']:5.1f}% "
                      f"{r['bound']:>4} {r['tflops']:8.1f} TF/s {r['gbs']:8.1f} GB/s frac {r['frac']:.3f}",
                      file=sys.stderr)
            for info, ms in per_op:
                if info.get("name") == "conv":
                    print(f"   {info['kernel']:<30} m={info['m']:8d} cout={info['cout']:5d} k={info['k']:5d} "
                          f"{1e3 * ms:9.1f} us {info['flops'] / ms / 1e9:8.1f} TF/s "
                          f"{info['bytes'] / ms / 1e6:8.1f} GB/s", file=sys.stderr)
                else:
                    print(f"   {info['kernel']:<30} {1e3 * ms:9.1f} us {info['bytes'] / ms / 1e6:8.1f} GB/s",
                          file=sys.stderr)
        result = {
            "metric": "frames/sec at 1024x1024 VToonify-D inference" if (4 * H, 4 * W) == (1024, 1024)
            and args.backbone == "dualstylegan" else f"frames/sec at {4 * W}x{4 * H} VToonify inference",
            "value": fps, "unit": "frames/s", "n_gpus": ws, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "timed_blocks": len(blocks), "timed_seconds": sum(blocks), "block_ms_per_step_min_max":
            [1e3 * min(blocks) / args.steps, 1e3 * max(blocks) / args.steps],
            "vs_baseline": None, "dtype": args.dtype, "data": "dry-run" if emu else "synthetic",
            "config": {"workload": f"VToonify-{'D' if args.backbone == 'dualstylegan' else 'T'} "
                                   f"22x{H}x{W} -> 3x{4 * H}x{4 * W}, batch {B} per GPU, d_s={d_s}, "
                                   f"seeded synthetic weights, style path recomputed every frame",
                       "frames_per_step_per_gpu": B, "parallelism": f"frame-parallel x{ws}",
                       "launch": "hipGraph replay" if use_graph else "eager",
                       "frames_in_flight_per_gpu": lanes, "tile_hints": args.tile_hints or None,
                       "plans": "per-image (VT_BATCH_EXACT=1)" if os.environ.get("VT_BATCH_EXACT") == "1" else "batch-aware",
                       "splitk_workgroup_target": 256,
                       "weight_broadcast_s": t_bcast, "host": host,
                       "per_rank_frames_per_s": fps / ws},
            "roofline": roofline,
            # the whole step against the chip: algorithmic FLOPs / bytes of SURVEY.md 8(d) (D 459.1 GFLOP, 2.21 GB bf16 per
            # 22x256x256 frame; T 400.5 / 1.87; x H*W/65536) over the wall time of the step -- a sanity bound (nothing skipped,
            # nothing above a peak), not the roofline claim, which is per kernel
            "end_to_end": end_to_end(args.backbone, H, W, B * ws, 1e-3 * (1e3 * elapsed / args.steps), args.dtype),
            "single_stream": single,
            "module_call": module_call,
            **extras,
            "output_checksum": checksum,
            "kernels": [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}
                        for r in rows[:8]],
        }
    # PCIe-inclusive rate of the same workload through the video driver (uint8 frames + parsing maps
    # in host memory -> uint8 frames in host memory; vtoonify_amd/video.py).  Reported beside `value`,
    # never as `value` (inputs of the timed region above are resident in HBM).
    if rank == 0 and ws == 1 and not args.no_video:
        result["pcie_inclusive"] = video_rate(eng, style, d_s, H, W, max(B, 4), use_graph)
        if not args.no_extras:
            try:
                result["pipeline"] = pipeline_rate(eng, style, d_s, H, W, max(B, 4), use_graph, dev)
            except Exception as e:   # never lose the bench line over an extra
                result["pipeline"] = {"error": f"{type(e).__name__}: {e}"}
    # the CPU baseline runs after the GPU numbers are final (rank 0, single-GPU runs only)
    if rank == 0 and ws == 1 and not args.no_cpu_baseline:
        del eng, sd_dev
        torch.cuda.empty_cache()
        result["cpu_baseline"] = cpu_baseline(args.backbone, H, W, args.cpu_budget)
    elif rank == 0:
        result["cpu_baseline"] = None
    if rank == 0:
        # the compact rates once more as the LAST key of the line (<= 1 KB): a reader who only sees the tail of the output
        # (the driver keeps 8 KB) still gets every configuration
        def _v(d, *path):
            for k in path:
                d = d.get(k) if isinstance(d, dict) else None
            return round(d, 1) if isinstance(d, (int, float)) else None
        r = result
        result["summary"] = {k: v for k, v in {
            "frames_per_s": round(r["value"], 1), "ms_per_step": round(r["ms_per_step"], 3), "single_stream": _v(r, "single_stream", "value"),
            "batch1": _v(r, "batch1", "value"), "batch16": _v(r, "batch16", "value"), "batch_exact": _v(r, "batch_exact", "value"),
            "config3": _v(r, "config3", "value"), "config4_T": _v(r, "config4", "T_1024x1024", "value"),
            "config4_D_d_s": (r.get("config4") or {}).get("D_d_s_sweep"),
            "config5_1536": _v(r, "config5", "D_1536x1536", "value"), "config5_1440x1600": _v(r, "config5", "D_1440x1600", "value"),
            "fp32": _v(r, "fp32", "value"), "fp32_batch": _v(r, "fp32", "headline_batch", "value"),
            "fp32x3": _v(r, "fp32x3", "value"), "fp32x3_batch": _v(r, "fp32x3", "headline_batch", "value"),
            "module_call": _v(r, "module_call", "value"), "pcie_inclusive": _v(r, "pcie_inclusive", "value"),
            "pipeline": _v(r, "pipeline", "value"), "cpu_baseline": (round(r["cpu_baseline"]["value"], 3), r["cpu_baseline"]["kind"])
            if r.get("cpu_baseline") else None,
            "roofline": [r["roofline"]["kernel"], round(r["roofline"]["avg_launch_us"], 1), round(r["roofline"]["frac"], 3)],
            "kernel_sum_ms": round(r["roofline"]["kernel_sum_ms_per_step"], 3), "launches": len(per_op) if rank == 0 else None,
        }.items() if v is not None}
        print(json.dumps(result))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
