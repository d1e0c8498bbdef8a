"""Frame-parallel sharding of one video across the GPUs of a node.

Inference has no cross-frame dependence: the style code is fixed per video
(style_transfer.py:138-150), the noise is zero (model/vtoonify.py:267) and InstanceNorm is
per sample.  So the video's frame index range is cut into `world_size` contiguous shards,
every rank keeps a full replica of the weights and runs the single-GPU path on its shard.
The ONLY collective on the path is a one-time broadcast from rank 0 of
  (i) the flattened state_dict (D: 166.6 M fp32 elements = 666 MB),
  (ii) the W+ style code (1,18,512) and the style degree,
over RCCL/xGMI (`backend="nccl"` on ROCm) -- or gloo in the CPU tests.  No per-frame
communication, no all-reduce.  Output frames stay on the rank that produced them (each
rank writes its own segment); `gather_frames` exists for callers that need a single
ordered writer.

One process per GPU; rendezvous via the usual env:// variables (RANK, LOCAL_RANK,
WORLD_SIZE, MASTER_ADDR, MASTER_PORT).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the environment (1-process defaults)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed when WORLD_SIZE > 1.  backend: "nccl" (= RCCL) on GPU
    ranks, "gloo" for CPU tests; default picks by device availability."""
    rank, local_rank, ws = world()
    # under a launcher (RANK set) the group is created even for one rank, so that the RCCL
    # broadcast path is the one exercised on a single-GPU box too
    if (ws > 1 or "RANK" in os.environ) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, init_method="env://", rank=rank, world_size=ws, **kw)
    return rank, local_rank, ws


def shard_range(n_frames: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous [start, stop) of the frames rank `rank` owns.  The first
    n_frames % world_size ranks get one extra frame; concatenating the shards in rank
    order reproduces the video order."""
    if n_frames < 0 or world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad shard request")
    base, extra = divmod(n_frames, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def batches(start: int, stop: int, batch_size: int) -> List[Tuple[int, int]]:
    """[start, stop) cut into batch_size pieces (last one ragged), as style_transfer.py:162
    accumulates `batch_size` frames before calling the model."""
    if batch_size < 1:
        raise ValueError("batch_size must be >= 1")
    return [(i, min(i + batch_size, stop)) for i in range(start, stop, batch_size)]


_UNUSED = None


def inference_unused(key: str) -> bool:
    """Is this state_dict entry one that VToonify.forward never reads?  The generator's 4x4 .. 32x32 layers
    (`conv1`, `convs.0-5`, `to_rgb1`, `to_rgbs.0-2`, the constant input: the frame enters at 32x32,
    model/vtoonify.py:245-250), the zero noise buffers (:267), the structure-transform linears of latent rows 0-6
    (`generator.res.0-6`: only rows 7-17 feed the synthesis convs, :221-224) and the FIR buffers of those layers.
    They are 68 M of the 166.6 M elements of the D checkpoint: the weight broadcast skips them (zeros of the right
    shape stand in, so the dict keeps the checkpoint's schema); tests/test_frames.py checks that zeroing them
    leaves the output bit-identical."""
    global _UNUSED
    if _UNUSED is None:
        import re
        _UNUSED = re.compile(r"^generator\.(generator\.)?(conv1\.|to_rgb1\.|input\.|noises\.|convs\.[0-5]\.|to_rgbs\.[0-2]\.)"
                             r"|^generator\.res\.[0-6]\.")
    return _UNUSED.match(key) is not None


def broadcast_state_dict(shapes: Dict[str, Tuple[int, ...]], sd: Optional[Dict[str, torch.Tensor]],
                         device: torch.device, src: int = 0,
                         bucket_elems: int = 64 << 20, skip_unused: bool = False) -> Dict[str, torch.Tensor]:
    """Broadcast an fp32 state_dict from `src` to every rank as a few large flat buckets
    (xGMI is point-to-point: few large messages, not 399 small ones).  `shapes` (key -> shape)
    is known on every rank (it is the checkpoint schema); only `src` needs `sd`.
    Returns tensors on `device`, views into the received buckets.  skip_unused: entries the frame path never
    reads (inference_unused) are not sent -- every rank gets zeros of their shape (D: 393 MB instead of 666 MB;
    the values that ARE sent stay fp32, so every rank packs / modulates exactly what a single GPU would)."""
    keys = sorted(k for k in shapes if not (skip_unused and inference_unused(k)))
    # every tensor starts on a 256-byte boundary of its bucket: the kernels' 16-byte vector paths (weights of the
    # style MLP, FIR taps ...) apply to the views exactly as they do to separately allocated parameters
    ALIGN = 64
    _padded = lambda k: (_numel(shapes[k]) + ALIGN - 1) // ALIGN * ALIGN
    ws = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    out: Dict[str, torch.Tensor] = {}
    i = 0
    while i < len(keys):
        j, n = i, 0
        while j < len(keys) and (n == 0 or n + _padded(keys[j]) <= bucket_elems):
            n += _padded(keys[j])
            j += 1
        flat = torch.zeros(n, dtype=torch.float32, device=device)
        if rank == src:
            off = 0
            for k in keys[i:j]:
                m = _numel(shapes[k])
                flat[off:off + m].copy_(sd[k].reshape(-1).to(torch.float32))
                off += _padded(k)
        if dist.is_initialized():
            dist.broadcast(flat, src=src)
        off = 0
        for k in keys[i:j]:
            m = _numel(shapes[k])
            out[k] = flat[off:off + m].view(shapes[k])
            off += _padded(k)
        i = j
    if skip_unused:
        for k in shapes:
            if k not in out:
                out[k] = torch.zeros(shapes[k], dtype=torch.float32, device=device)
    return out


def broadcast_style(style: Optional[torch.Tensor], d_s: Optional[float], device: torch.device,
                    src: int = 0) -> Tuple[torch.Tensor, float]:
    """W+ style code (1,18,512) + style degree, one 36 KB message."""
    buf = torch.empty(18 * 512 + 1, dtype=torch.float32, device=device)
    rank = dist.get_rank() if dist.is_initialized() else 0
    if rank == src:
        buf[:-1].copy_(style.reshape(-1).to(torch.float32))
        buf[-1] = float(d_s)
    if dist.is_initialized():
        dist.broadcast(buf, src=src)
    return buf[:-1].view(1, 18, 512).clone(), float(buf[-1].item())


def gather_frames(local: torch.Tensor, n_frames: int, dst: int = 0) -> Optional[torch.Tensor]:
    """Ordered gather of per-rank output frames (uint8 or float, (n_local, 3, H, W)) to `dst`.
    Shards may differ by one frame, so each rank pads to the largest shard."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    ws, rank = dist.get_world_size(), dist.get_rank()
    counts = [shard_range(n_frames, r, ws) for r in range(ws)]
    most = max(b - a for a, b in counts)
    pad = torch.zeros((most,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]].copy_(local)
    got = [torch.empty_like(pad) for _ in range(ws)] if rank == dst else None
    dist.gather(pad, got, dst=dst)
    if rank != dst:
        return None
    return torch.cat([g[:b - a] for g, (a, b) in zip(got, counts)], 0)


def _numel(shape) -> int:
    n = 1
    for s in shape:
        n *= int(s)
    return n
