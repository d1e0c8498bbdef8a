#!/usr/bin/env python3
"""`style_transfer.py` on MI355X: the reference's command line over the frame-parallel video driver.

    python tools/style_transfer_amd.py --content clip.mp4 --video --scale_image --style_id 26 --style_degree 0.5 \
        --ckpt checkpoint/vtoonify_d_cartoon/vtoonify_s_d.pt --batch_size 4
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        tools/style_transfer_amd.py --content clip.npy --video --parsing_map_path maps.npy ...

The 16 options of the reference (style_transfer.py:17-46) keep their names, types, defaults and meaning; the loop is the
reference's (:99-183) with its host work moved to the GPU (vtoonify_amd/video.py) and its frames cut into one contiguous
shard per rank (vtoonify_amd/frames.py: rank 0 reads the checkpoint and the style code, ONE bucketed RCCL broadcast
hands them to the other ranks, no per-frame communication; the output is written in frame order).

Frame I/O is pluggable because the reference's readers are optional third-party packages (cv2, dlib) that an MI355X
serving image need not carry:
  content   *.mp4 / *.avi / *.jpg / *.png  -> cv2 (when importable), frames BGR as VideoCapture.read() delivers them
            *.npy                          -> (N,H,W,3) uint8 array (memory-mapped), channel order --frame_order
            a directory                    -> sorted *.npy frames (H,W,3) uint8
  output    the same kind as the content: `<basename>_vtoonify_<d|t>.mp4|.npy|/` under --output_path
`--scale_image` (FaceCrop below: the reference's resize + crop from the first frame's eye distance) and the aligned style
crop need the face landmarks of dlib and cv2's filter / resize (util.py:163-188, model/encoder/align_all_parallel.py);
without them pass pre-cropped frames, and either `--intrinsic_code` (the pSp encoder's (1,18,512) output, .npy) or accept
the un-aligned first frame as the style encoder's input (a warning is printed).
`--ckpt synthetic` / `--style_encoder_path synthetic` / `--faceparsing_path synthetic` build seeded random weights of the
reference's schema (there are no checkpoints on the benchmark boxes); everything else is the reference's behaviour.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from vtoonify_amd import _lib, frames, synth  # noqa: E402
from vtoonify_amd.video import VideoToonifier, frame_pack  # noqa: E402
from vtoonify_amd.vtoonify import VToonify  # noqa: E402

VIDEO_EXT = (".mp4", ".avi", ".mov", ".mkv", ".webm")
IMAGE_EXT = (".jpg", ".jpeg", ".png", ".bmp")


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Style Transfer")
    # ---- the reference's options, verbatim (style_transfer.py:21-37) ----
    p.add_argument("--content", type=str, default="./data/077436.jpg", help="path of the content image/video")
    p.add_argument("--style_id", type=int, default=26, help="the id of the style image")
    p.add_argument("--style_degree", type=float, default=0.5, help="style degree for VToonify-D")
    p.add_argument("--color_transfer", action="store_true", help="transfer the color of the style")
    p.add_argument("--ckpt", type=str, default="./checkpoint/vtoonify_d_cartoon/vtoonify_s_d.pt", help="path of the saved model")
    p.add_argument("--output_path", type=str, default="./output/", help="path of the output images")
    p.add_argument("--scale_image", action="store_true", help="resize and crop the image to best fit the model")
    p.add_argument("--style_encoder_path", type=str, default="./checkpoint/encoder.pt", help="path of the style encoder")
    p.add_argument("--exstyle_path", type=str, default=None, help="path of the extrinsic style code")
    p.add_argument("--faceparsing_path", type=str, default="./checkpoint/faceparsing.pth", help="path of the face parsing model")
    p.add_argument("--video", action="store_true", help="if true, video stylization; if false, image stylization")
    p.add_argument("--cpu", action="store_true", help="if true, only use cpu")
    p.add_argument("--backbone", type=str, default="dualstylegan", help="dualstylegan | toonify")
    p.add_argument("--padding", type=int, nargs=4, default=[200, 200, 200, 200],
                   help="left, right, top, bottom paddings to the face center")
    p.add_argument("--batch_size", type=int, default=4, help="batch size of frames when processing video")
    p.add_argument("--parsing_map_path", type=str, default=None, help="path of the refined parsing map of the target video")
    # ---- additions of this driver ----
    p.add_argument("--precision", choices=["fp32", "fp32_exact", "bf16"], default=None,
                   help="arithmetic of the frame (default: VTOONIFY_AMD_DTYPE or fp32 = the reference's precision)")
    p.add_argument("--depth", type=int, default=3, help="batches in flight per GPU (each on its own stream and plan)")
    p.add_argument("--frame_order", choices=["bgr", "rgb"], default="bgr", help="channel order of .npy frames (cv2 files are BGR)")
    p.add_argument("--intrinsic_code", type=str, default=None,
                   help=".npy (1,18,512): the style encoder's output for this video (skips the pSp pass and the face alignment)")
    p.add_argument("--max_frames", type=int, default=None, help="stop after this many frames")
    p.add_argument("--seed", type=int, default=0, help="seed of `synthetic` weights")
    return p


def parse(argv=None):
    opt = build_parser().parse_args(argv)
    if opt.exstyle_path is None:                         # style_transfer.py:41-42
        opt.exstyle_path = os.path.join(os.path.dirname(opt.ckpt), "exstyle_code.npy")
    return opt


# ----------------------------------------------------------------------------------------- frame sources / sinks
class NpySource:
    """(N,H,W,3) uint8 .npy, memory-mapped; random access, so every rank reads only its shard."""
    kind = "npy"

    def __init__(self, path, bgr):
        self.a = np.load(path, mmap_mode="r")
        if self.a.ndim == 3:
            self.a = self.a[None]
        if self.a.ndim != 4 or self.a.shape[3] != 3 or self.a.dtype != np.uint8:
            raise ValueError(f"{path}: expected (N,H,W,3) uint8 frames")
        self.bgr, self.fps = bgr, 25.0

    def __len__(self):
        return self.a.shape[0]

    def frames(self, start, stop):
        for i in range(start, stop):
            yield np.ascontiguousarray(self.a[i])


class DirSource:
    kind = "dir"

    def __init__(self, path, bgr):
        self.files = sorted(os.path.join(path, f) for f in os.listdir(path) if f.endswith(".npy"))
        if not self.files:
            raise ValueError(f"{path}: no *.npy frames")
        self.bgr, self.fps = bgr, 25.0

    def __len__(self):
        return len(self.files)

    def frames(self, start, stop):
        for f in self.files[start:stop]:
            yield np.ascontiguousarray(np.load(f))


class Cv2Source:
    """cv2.VideoCapture / cv2.imread: BGR frames, as the reference reads them (style_transfer.py:103-112,188)."""

    def __init__(self, path, video):
        import cv2
        self.cv2, self.path, self.video, self.bgr = cv2, path, video, True
        self.kind = "video" if video else "image"
        if video:
            cap = cv2.VideoCapture(path)
            self.n, self.fps = int(cap.get(7)), cap.get(5)
            cap.release()
        else:
            self.n, self.fps = 1, 25.0

    def __len__(self):
        return self.n

    def frames(self, start, stop):
        if not self.video:
            yield self.cv2.imread(self.path)
            return
        cap = self.cv2.VideoCapture(self.path)
        cap.set(self.cv2.CAP_PROP_POS_FRAMES, start)
        for _ in range(start, stop):
            ok, fr = cap.read()
            if not ok:
                break
            yield fr
        cap.release()


def open_source(path, video, frame_order):
    ext = os.path.splitext(path)[1].lower()
    if os.path.isdir(path):
        return DirSource(path, frame_order == "bgr")
    if ext == ".npy":
        return NpySource(path, frame_order == "bgr")
    if ext in VIDEO_EXT + IMAGE_EXT:
        try:
            import cv2  # noqa: F401
        except ImportError:
            raise SystemExit(f"{path}: reading {ext} needs cv2 (not importable here); pass frames as .npy "
                             "((N,H,W,3) uint8) or a directory of .npy frames") from None
        return Cv2Source(path, video and ext in VIDEO_EXT)
    raise SystemExit(f"{path}: unknown content type")


class NpySink:
    """One (N,4H,4W,3) uint8 .npy; rank 0 creates it, every rank writes its own rows: the file is in frame order without
    a gather (one node, one file system)."""

    def __init__(self, path, n, h, w, rank, barrier):
        self.path = path
        if rank == 0:
            np.lib.format.open_memmap(path, mode="w+", dtype=np.uint8, shape=(n, h, w, 3)).flush()
        barrier()
        self.a = np.load(path, mmap_mode="r+")

    def __call__(self, i, frame):
        self.a[i] = frame

    def close(self):
        self.a.flush()
        del self.a


class DirSink:
    def __init__(self, path, rank, barrier):
        self.path = path
        if rank == 0:
            os.makedirs(path, exist_ok=True)
        barrier()

    def __call__(self, i, frame):
        np.save(os.path.join(self.path, f"{i:06d}.npy"), frame)

    def close(self):
        pass


class Cv2Sink:
    """cv2.VideoWriter / imwrite on rank 0 (style_transfer.py:128-130,181): the other ranks' frames arrive by ordered gather."""
    needs_gather = True

    def __init__(self, path, fps, h, w, video):
        import cv2
        self.cv2, self.path, self.video = cv2, path, video
        self.w = cv2.VideoWriter(path, cv2.VideoWriter_fourcc(*"mp4v"), fps, (w, h)) if video else None

    def __call__(self, i, frame):
        if self.video:
            self.w.write(frame)
        else:
            self.cv2.imwrite(self.path, frame)

    def close(self):
        if self.w is not None:
            self.w.release()


class FaceCrop:
    """--scale_image (style_transfer.py:113-127,150-155): the first frame's eye distance fixes one resize + crop for the whole
    video (util.py:163-188: 64 pixels between the eyes, --padding around their centre, multiples of 8); frames of a high-
    resolution source are low-pass filtered first ([1,3,3,1]/8 on both axes, once at scale <= 0.75, twice at <= 0.375).  The
    landmarks are dlib's and the filter / resize are cv2's, exactly the calls the reference makes: without the two packages
    this option stops with a message instead of approximating them."""

    def __init__(self, frame, bgr, padding):
        try:
            import cv2
            import dlib
        except ImportError:
            raise SystemExit("--scale_image needs cv2 and dlib (face landmarks, util.py:get_video_crop_parameter); they are "
                             "not importable here: pass frames that are already cropped") from None
        from model.encoder.align_all_parallel import get_landmark        # the reference's helper, through the import mirror
        self.cv2 = cv2
        predictor = dlib.shape_predictor("./checkpoint/shape_predictor_68_face_landmarks.dat")
        rgb = np.ascontiguousarray(frame[..., ::-1] if bgr else frame)
        lm = get_landmark(rgb, predictor)
        if lm is None:
            raise SystemExit("--scale_image: no face found in the first frame")
        eye_l, eye_r = lm[36:42], lm[42:48]
        self.scale = scale = 64.0 / (np.mean(eye_r[:, 0]) - np.mean(eye_l[:, 0]))
        cx, cy = ((np.mean(eye_r, axis=0) + np.mean(eye_l, axis=0)) / 2) * scale
        self.h, self.w = round(frame.shape[0] * scale), round(frame.shape[1] * scale)
        self.left = max(round(cx - padding[0]), 0) // 8 * 8
        self.right = min(round(cx + padding[1]), self.w) // 8 * 8
        self.top = max(round(cy - padding[2]), 0) // 8 * 8
        self.bottom = min(round(cy + padding[3]), self.h) // 8 * 8
        self.k = np.array([[0.125], [0.375], [0.375], [0.125]])

    def __call__(self, frame):
        cv2 = self.cv2
        if self.scale <= 0.75:
            frame = cv2.sepFilter2D(frame, -1, self.k, self.k)
        if self.scale <= 0.375:
            frame = cv2.sepFilter2D(frame, -1, self.k, self.k)
        return cv2.resize(frame, (self.w, self.h))[self.top:self.bottom, self.left:self.right]


# ----------------------------------------------------------------------------------------- weights and style
def _shapes(tag):
    with open(os.path.join(REPO, "tests", "golden", f"keys_{tag}.json")) as f:
        return {k: tuple(v) for k, v in json.load(f).items()}


def load_generator_weights(opt, model):
    if opt.ckpt.startswith("synthetic"):
        return synth.synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, opt.seed)
    return torch.load(opt.ckpt, map_location="cpu")["g_ema"]            # style_transfer.py:63


def style_code(opt, model, first_frame, bgr, device, log):
    """W+ code of the video (style_transfer.py:135-147, 76-81): pSp on the (aligned) first frame -> zplus2wplus -> rows 0..6 from
    the extrinsic style (or all 18 with --color_transfer)."""
    if opt.intrinsic_code:
        z = torch.from_numpy(np.load(opt.intrinsic_code)).float().reshape(1, 18, 512).to(device)
    else:
        from vtoonify_amd.psp import GradualStyleEncoder
        face = first_frame[..., ::-1] if bgr else first_frame                       # RGB
        try:
            import dlib  # noqa: F401
            from model.encoder.align_all_parallel import align_face                 # the reference's own alignment, via the mirror
            lm = dlib.shape_predictor("./checkpoint/shape_predictor_68_face_landmarks.dat")
            face = np.asarray(align_face(np.ascontiguousarray(face), lm))
        except ImportError:
            log("[style] dlib / align_face not importable: the style encoder sees the un-aligned first frame resized to 256x256")
        t = torch.from_numpy(np.ascontiguousarray(face)).to(device).permute(2, 0, 1)[None].float() / 255.0
        t = torch.nn.functional.interpolate((t - 0.5) / 0.5, size=(256, 256), mode="bilinear", align_corners=False)
        dt = torch.float32 if device.type == "cpu" else torch.bfloat16
        psp = GradualStyleEncoder(50, "ir_se", compute_dtype=dt)
        if opt.style_encoder_path.startswith("synthetic"):
            psp.load_state_dict(synth.synth_state_dict(_shapes("psp"), opt.seed))
            latent_avg = torch.zeros(18, 512)
        else:
            ck = torch.load(opt.style_encoder_path, map_location="cpu")               # util.py:150-160
            psp.load_state_dict({k[len("encoder."):]: v for k, v in ck["state_dict"].items() if k.startswith("encoder.")})
            latent_avg = ck["latent_avg"]
        psp.eval().to(device)
        with torch.no_grad():
            z = psp(t).float() + latent_avg.to(device).reshape(1, -1, 512)
    with torch.no_grad():
        s_w = model.zplus2wplus(z).clone()
        if model.backbone == "dualstylegan":
            if opt.ckpt.startswith("synthetic") and not os.path.exists(opt.exstyle_path):
                ex = synth.synth_style(seed=100 + opt.style_id).to(device)
            else:
                exstyles = np.load(opt.exstyle_path, allow_pickle=True).item()
                ex = torch.tensor(exstyles[list(exstyles.keys())[opt.style_id]]).to(device)
            ex = model.zplus2wplus(ex.reshape(1, 18, 512).float())
            if opt.color_transfer:
                s_w = ex
            else:
                s_w[:, :7] = ex[:, :7]
    return s_w


def parsing_engine(opt, device, dtype):
    from vtoonify_amd.bisenet import BiSeNet
    net = BiSeNet(n_classes=19, compute_dtype=dtype)
    if opt.faceparsing_path.startswith("synthetic"):
        net.load_state_dict(synth.synth_state_dict(_shapes("bisenet"), opt.seed))
    else:
        net.load_state_dict(torch.load(opt.faceparsing_path, map_location="cpu"))
    return net.to(device).eval()


# ----------------------------------------------------------------------------------------- main
def main(argv=None, device=None, backend=None) -> dict:
    """Returns a small report (frames, seconds, output path).  `device` / `backend` are for tests (host emulation + gloo)."""
    opt = parse(argv)
    rank, local_rank, ws = frames.init(backend)
    import torch.distributed as dist
    barrier = dist.barrier if dist.is_initialized() else (lambda: None)
    log = (lambda *a: print(*a, flush=True)) if rank == 0 else (lambda *a: None)
    if rank == 0:
        log("Load options")
        for k, v in sorted(vars(opt).items()):
            log(f"{k}: {v}")
        log("*" * 98)
    if device is None:
        device = torch.device("cpu") if opt.cpu else torch.device("cuda", local_rank)
    device = torch.device(device)
    if device.type == "cuda":
        torch.cuda.set_device(device)
    on_engine = device.type == "cuda" or _lib.emulation_injected()
    prec = opt.precision or os.environ.get("VTOONIFY_AMD_DTYPE", "fp32")
    cdt = torch.bfloat16 if prec == "bf16" else torch.float32

    src = open_source(opt.content, opt.video, opt.frame_order)
    n = len(src) if opt.max_frames is None else min(len(src), opt.max_frames)
    if not on_engine and not (opt.video and opt.parsing_map_path):
        raise SystemExit("--cpu needs --parsing_map_path: the parsing network runs on the GPU only")

    # ---- weights: rank 0 reads the checkpoint, one bucketed broadcast (RCCL over xGMI) hands it to the other ranks ----
    model = VToonify(backbone=opt.backbone, compute_dtype=cdt, exact_fp32=(prec == "fp32_exact"))
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = load_generator_weights(opt, model) if rank == 0 else None
    if ws > 1:
        sd = frames.broadcast_state_dict(shapes, sd, device, skip_unused=True)
    model.load_state_dict(sd)
    model.to(device).eval()

    # ---- style code: once per video, on rank 0 ----
    first = next(iter(src.frames(0, 1)))
    crop = FaceCrop(first, src.bgr, opt.padding) if opt.scale_image else None       # parameters of the FIRST frame, for all
    if crop is not None:
        first = crop(first)
    H, W = first.shape[0] // 8 * 8, first.shape[1] // 8 * 8                       # util.py:184-187 crops to //8*8
    s_w = style_code(opt, model, first, src.bgr, device, log) if rank == 0 else None
    d_s = opt.style_degree if opt.backbone == "dualstylegan" else None
    if ws > 1:
        s_w, d = frames.broadcast_style(s_w, opt.style_degree, device)
        d_s = d if opt.backbone == "dualstylegan" else None
    log("Load models successfully!")

    # ---- parsing maps: given (--parsing_map_path, style_transfer.py:168-169) or computed on the GPU (:170-172) ----
    maps = np.load(opt.parsing_map_path, mmap_mode="r") if (opt.video and opt.parsing_map_path) else None
    par = None if maps is not None else parsing_engine(opt, device, cdt)

    base = os.path.basename(opt.content.rstrip("/")).split(".")[0]
    stem = os.path.join(opt.output_path, f"{base}_vtoonify_{opt.backbone[0]}")
    if rank == 0:
        os.makedirs(opt.output_path, exist_ok=True)
    a, b = frames.shard_range(n, rank, ws)
    gather = False
    if src.kind == "npy":
        out_path = stem + ".npy"
        sink = NpySink(out_path, n, 4 * H, 4 * W, rank, barrier)
    elif src.kind == "dir":
        out_path = stem
        sink = DirSink(out_path, rank, barrier)
    else:
        out_path = stem + (".mp4" if src.kind == "video" else ".jpg")
        gather = ws > 1
        sink = Cv2Sink(out_path, src.fps, 4 * H, 4 * W, src.kind == "video") if rank == 0 else None
    log(f"Processing {os.path.basename(opt.content)} with vtoonify_{opt.backbone[0]}: {n} frames, {ws} rank(s), "
        f"{opt.batch_size} per batch, precision {model.precision if on_engine else 'fp32 (torch, CPU)'}")

    def shard_source():
        for j, fr in enumerate(src.frames(a, b)):
            fr = (fr if crop is None else crop(fr))[:H, :W]
            yield fr, (None if maps is None else np.asarray(maps[a + j], dtype=np.float32)[:, :H, :W])

    t0 = time.time()
    local = [] if gather else None
    emit = (lambda i, fr: local.append(fr.copy())) if gather else sink
    if on_engine:
        vt = VideoToonifier(model.engine(), s_w, d_s, batch_size=opt.batch_size, bgr=src.bgr, depth=opt.depth,
                            parsing_engine=None if par is None else par.engine())
        done = vt.run(shard_source(), emit, first_index=a)
        if device.type == "cuda":
            torch.cuda.synchronize(device)
    else:
        done = _cpu_loop(opt, model, par, s_w, d_s, shard_source(), emit, a, src.bgr)     # the reference's loop, on torch
    if gather:
        t = torch.from_numpy(np.stack(local, 0)) if local else torch.zeros((0, 4 * H, 4 * W, 3), dtype=torch.uint8)
        allf = frames.gather_frames(t.to(device) if dist.get_backend() == "nccl" else t, n)
        if rank == 0:
            for i, fr in enumerate(allf.cpu().numpy()):
                sink(i, fr)
    if sink is not None:
        sink.close()
    barrier()
    dt = time.time() - t0
    log(f"Transfer style successfully!  {n} frames in {dt:.2f} s ({n / max(dt, 1e-9):.1f} frames/s incl. I/O) -> {out_path}")
    if dist.is_initialized() and "RANK" in os.environ and backend is None:   # launched by torch.distributed.run: leave the group cleanly
        dist.destroy_process_group()
    return {"frames": n, "shard": (a, b), "done": done, "seconds": dt, "output": out_path, "rank": rank, "world_size": ws}


def _cpu_loop(opt, model, par, s_w, d_s, source, emit, first_index, bgr):
    """`--cpu` (style_transfer.py:32,55): the reference's per-batch sequence on CPU tensors -- the module runs the eager graph
    over the operator surface's CPU branch (vtoonify_amd/eager.py)."""
    idx, batch = first_index, []

    def flush():
        nonlocal idx, batch
        if not batch:
            return
        fr = np.stack([f for f, _ in batch], 0)
        rgb = fr[..., ::-1] if bgr else fr
        x = (torch.from_numpy(np.ascontiguousarray(rgb)).permute(0, 3, 1, 2).float() / 255.0 - 0.5) / 0.5
        if batch[0][1] is not None:
            x_p = torch.from_numpy(np.stack([p for _, p in batch], 0))
        else:
            x_p = par.parsing_maps(x)
        with torch.no_grad():
            y = model(torch.cat((x, x_p / 16.0), 1), s_w.repeat(x.shape[0], 1, 1), d_s=d_s).clamp(-1, 1)
        out = ((y.permute(0, 2, 3, 1) + 1.0) * 127.5).numpy().astype(np.uint8)        # tensor2cv2, util.py:190-192
        for o in out:
            emit(idx, np.ascontiguousarray(o[..., ::-1] if bgr else o))
            idx += 1
        batch = []

    for item in source:
        batch.append(item)
        if len(batch) == opt.batch_size:
            flush()
    flush()
    return idx - first_index


if __name__ == "__main__":
    main()
