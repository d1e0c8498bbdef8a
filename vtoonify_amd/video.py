"""Frame-parallel video driver around the whole-frame engine.

What the reference does per batch on the host (style_transfer.py:99-183) and what happens here:

  reference (CPU, synchronous)                         here
  ---------------------------------------------------  -----------------------------------------
  cv2.cvtColor(BGR2RGB); transform(frame).to(device)   uint8 frames -> pinned buffer -> H2D on a
    (ToTensor+Normalize on the CPU, fp32 H2D: 4x the     copy stream; vt_frame_pack on the GPU
    bytes)                                               (swap + normalise + cat with x_p/16)
  vtoonify(inputs, s_w.repeat(B,1,1), d_s)             VToonifyEngine.forward (hipGraph replay)
  torch.clamp; tensor2cv2(y[k].cpu()) per frame        vt_frame_unpack on the GPU (clamp, *127.5,
    (12.6 MB fp32 D2H per 1024^2 frame, numpy on CPU)    uint8, BGR): 3 MB D2H on the copy stream
  videoWriter2.write(frame)                            sink(index, frame) in frame order

`depth` batches are in flight: while batch k computes, batch k+1 is staged/uploaded and batch
k-1 is downloaded and handed to the sink.  Every in-flight batch has its own compute stream and
its own engine lane (plan buffers + hipGraph): the kernels of consecutive batches overlap on the
GPU, which is worth +37 % (2 in flight) / +50 % (3) frames/s at one frame per batch, where most
launches of a frame are too small to fill 256 CUs on their own.  One process per GPU; a multi-GPU job cuts the
frame range with frames.shard_range and every rank runs this loop on its shard (no per-frame
communication).  The parsing maps are either an input (the reference's --parsing_map_path branch,
style_transfer.py:168-169) or, when the source yields None for them and a `parsing_engine`
(vtoonify_amd.bisenet.BiSeNetEngine) is given, computed on the GPU from the frame itself
(style_transfer.py:170-172) in the same compute stream, before the frame is packed.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib, frames as _frames


def _ptr(t: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(t.data_ptr())


def frame_pack(frames_u8: torch.Tensor, parsing: Optional[torch.Tensor] = None, bgr: bool = True,
               parsing_scale: float = 1.0 / 16.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(N,H,W,3) uint8 [+ (N,pc,H,W) fp32] -> (N,3+pc,H,W) fp32 network input (vt_frame_pack)."""
    if frames_u8.dtype != torch.uint8 or frames_u8.ndim != 4 or frames_u8.shape[3] != 3:
        raise _lib.VtError("frame_pack: frames must be (N,H,W,3) uint8")
    n, h, w, _ = frames_u8.shape
    pc = 0
    if parsing is not None:
        if parsing.dtype != torch.float32 or parsing.ndim != 4 or parsing.shape[0] != n or tuple(parsing.shape[2:]) != (h, w):
            raise _lib.VtError("frame_pack: parsing must be (N,pc,H,W) float32 at the frame size")
        pc = parsing.shape[1]
        parsing = parsing.contiguous()
    frames_u8 = frames_u8.contiguous()
    if out is None:
        out = torch.empty((n, 3 + pc, h, w), dtype=torch.float32, device=frames_u8.device)
    _lib.check(_lib.lib().vt_frame_pack(_ptr(out), _ptr(frames_u8), int(bgr), _ptr(parsing) if pc else None, pc,
                                        float(parsing_scale), n, h, w, _stream(frames_u8)), "vt_frame_pack")
    return out


def frame_unpack(image: torch.Tensor, bgr: bool = True, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(N,3,H,W) fp32 -> (N,H,W,3) uint8, clamp + tensor2cv2 (vt_frame_unpack)."""
    if image.dtype != torch.float32 or image.ndim != 4 or image.shape[1] != 3:
        raise _lib.VtError("frame_unpack: image must be (N,3,H,W) float32")
    image = image.contiguous()
    n, _, h, w = image.shape
    if out is None:
        out = torch.empty((n, h, w, 3), dtype=torch.uint8, device=image.device)
    _lib.check(_lib.lib().vt_frame_unpack(_ptr(out), _ptr(image), int(bgr), n, h, w, _stream(image)), "vt_frame_unpack")
    return out


def _stream(t: torch.Tensor):
    if t.device.type == "cuda":
        return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return C.c_void_p(0)


class _Slot:
    """Staging buffers of one in-flight batch."""

    def __init__(self, B, H, W, pc, device, lane=0, host_parsing=True):
        cuda = device.type == "cuda"
        self.lane = lane
        self.compute = torch.cuda.Stream(device) if cuda else None
        self.h_frames = torch.empty((B, H, W, 3), dtype=torch.uint8, pin_memory=cuda)
        self.h_parsing = (torch.empty((B, pc, H, W), dtype=torch.float32, pin_memory=cuda)
                          if pc and host_parsing else None)
        self.d_rgb = torch.empty((B, 3, H, W), dtype=torch.float32, device=device) if pc and not host_parsing else None
        self.h_out = torch.empty((B, 4 * H, 4 * W, 3), dtype=torch.uint8, pin_memory=cuda)
        # zero-filled: a ragged final batch runs the full-B plan over whatever the unused rows hold
        self.d_frames = torch.zeros((B, H, W, 3), dtype=torch.uint8, device=device)
        self.d_parsing = torch.zeros((B, pc, H, W), dtype=torch.float32, device=device) if pc else None
        self.d_x = torch.empty((B, 3 + pc, H, W), dtype=torch.float32, device=device)
        self.d_out = torch.empty((B, 4 * H, 4 * W, 3), dtype=torch.uint8, device=device)
        self.ev_up = torch.cuda.Event() if cuda else None
        self.ev_done = torch.cuda.Event() if cuda else None
        self.ev_down = torch.cuda.Event() if cuda else None
        # numpy views of the pinned staging buffers: frames are staged with plain memcpy (np.copyto);
        # torch's CPU copy spreads a 5 MB copy over the intra-op thread pool, whose wake-up after the
        # main thread has slept in an event wait costs milliseconds per frame
        self.n_frames = self.h_frames.numpy()
        self.n_parsing = self.h_parsing.numpy() if self.h_parsing is not None else None
        self.n_out = self.h_out.numpy()
        self.count = 0
        self.first = 0


class VideoToonifier:
    """engine: VToonifyEngine (or anything with .forward(x, style, d_s, shared_style=, use_graph=)
    and .device).  style: W+ (1,18,512) for the whole video (style_transfer.py:138-150), d_s the
    style degree.  Frames are (H,W,3) uint8, BGR when `bgr` (cv2 order) else RGB; output frames
    are (4H,4W,3) uint8 in the same channel order."""

    def __init__(self, engine, style: torch.Tensor, d_s: Optional[float], batch_size: int = 4, bgr: bool = True,
                 depth: int = 2, use_graph: bool = True, parsing_engine=None, parsing_channels: int = 19):
        if batch_size < 1 or depth < 1:
            raise ValueError("batch_size and depth must be >= 1")
        self.engine, self.style, self.d_s = engine, style, d_s
        self.B, self.bgr, self.depth = batch_size, bgr, depth
        self.device = engine.device
        self.cuda = self.device.type == "cuda"
        self.use_graph = use_graph and self.cuda
        self.parsing_engine, self.parsing_channels = parsing_engine, parsing_channels
        self._slots: List[_Slot] = []
        self._geom = None
        if self.cuda:
            # uploads and downloads on separate streams: the download of batch k waits for its
            # compute, and an upload queued behind it would hold back batch k+1's compute with it
            self.copy_up = torch.cuda.Stream(self.device)
            self.copy_down = torch.cuda.Stream(self.device)

    # -- one batch through the three stages -------------------------------------------------
    def _slots_for(self, H, W, pc, host_parsing=True):
        if self._geom != (H, W, pc, host_parsing):
            if H % 8 or W % 8:
                raise _lib.VtError("frame height and width must be multiples of 8 (util.py:184-187)")
            self._slots = [_Slot(self.B, H, W, pc, self.device, lane=i, host_parsing=host_parsing)
                           for i in range(self.depth)]
            self._geom = (H, W, pc, host_parsing)
        return self._slots

    def _submit(self, slot: _Slot, n: int):
        """Upload, compute, download `n` staged frames of `slot` (asynchronous on CUDA)."""
        pc = 0 if slot.d_parsing is None else slot.d_parsing.shape[1]
        host_p = slot.h_parsing is not None
        if self.cuda:
            with torch.cuda.stream(self.copy_up):
                slot.d_frames[:n].copy_(slot.h_frames[:n], non_blocking=True)
                if host_p:
                    slot.d_parsing[:n].copy_(slot.h_parsing[:n], non_blocking=True)
                slot.ev_up.record(self.copy_up)
            with torch.cuda.stream(slot.compute):
                slot.compute.wait_event(slot.ev_up)
                self._compute(slot, n, pc)
                slot.ev_done.record(slot.compute)
            with torch.cuda.stream(self.copy_down):
                self.copy_down.wait_event(slot.ev_done)
                slot.h_out[:n].copy_(slot.d_out[:n], non_blocking=True)
                slot.ev_down.record(self.copy_down)
        else:
            slot.d_frames[:n].copy_(slot.h_frames[:n])
            if host_p:
                slot.d_parsing[:n].copy_(slot.h_parsing[:n])
            self._compute(slot, n, pc)
            slot.h_out[:n].copy_(slot.d_out[:n])

    def _compute(self, slot: _Slot, n: int, pc: int):
        # a ragged final batch (n < B) still runs the B-frame plan: one set of plan buffers and one hipGraph
        # per lane, no capture in the middle of the pipeline; rows n..B-1 are stale and never retired
        n = self.B
        if slot.d_rgb is not None:
            # no parsing maps from the source: x_p = nearest_x0.5(BiSeNet(2 * bilinear_x2(x))[0]) on the
            # GPU (style_transfer.py:170-172); the /16 is vt_frame_pack's parsing_scale
            rgb = frame_pack(slot.d_frames[:n], None, self.bgr, out=slot.d_rgb[:n])
            self.parsing_engine.parsing_maps(rgb, use_graph=self.use_graph, lane=slot.lane, out=slot.d_parsing[:n])
        x = frame_pack(slot.d_frames[:n], slot.d_parsing[:n] if pc else None, self.bgr, out=slot.d_x[:n])
        # the frame is converted to uint8 by the next launch on this stream: borrow the engine's output buffer
        kw = {"borrow": True} if getattr(self.engine, "supports_borrow", False) else {}
        y = self.engine.forward(x, self.style, self.d_s, shared_style=True, use_graph=self.use_graph,
                                lane=slot.lane, **kw)
        frame_unpack(y, self.bgr, out=slot.d_out[:n])

    # -- public -----------------------------------------------------------------------------
    def run(self, source: Iterable[Tuple[np.ndarray, Optional[np.ndarray]]],
            sink: Callable[[int, np.ndarray], None], first_index: int = 0) -> int:
        """source yields (frame (H,W,3) uint8, parsing (pc,H,W) float32 or None) in frame order;
        sink(index, frame (4H,4W,3) uint8) is called in frame order (the returned array is only
        valid during the call -- it is a view of a recycled pinned buffer).  Returns the number
        of frames processed."""
        pending: List[_Slot] = []
        it: Iterator = iter(source)
        idx = first_index
        k = 0
        done = False
        while not done or pending:
            if not done:
                batch = []
                for item in it:
                    batch.append(item)
                    if len(batch) == self.B:
                        break
                if len(batch) < self.B:
                    done = True
                if batch:
                    f0, p0 = batch[0]
                    host_p = p0 is not None or self.parsing_engine is None
                    pc = (0 if p0 is None else p0.shape[0]) if host_p else self.parsing_channels
                    slots = self._slots_for(f0.shape[0], f0.shape[1], pc, host_p)
                    slot = slots[k % self.depth]
                    k += 1
                    if slot in pending:   # the ring is full: retire the oldest batch first
                        self._retire(pending.pop(0), sink)
                    for j, (f, p) in enumerate(batch):
                        if f.shape != f0.shape or f.dtype != np.uint8:
                            raise _lib.VtError("all frames of a video must share one (H,W,3) uint8 shape")
                        np.copyto(slot.n_frames[j], f)
                        if pc and host_p:
                            np.copyto(slot.n_parsing[j], p, casting="same_kind")
                    slot.count, slot.first = len(batch), idx
                    idx += len(batch)
                    self._submit(slot, slot.count)
                    pending.append(slot)
                    continue
            if pending:
                self._retire(pending.pop(0), sink)
        return idx - first_index

    def _retire(self, slot: _Slot, sink):
        if self.cuda:
            slot.ev_down.synchronize()
        for j in range(slot.count):
            sink(slot.first + j, slot.n_out[j])


def toonify_shard(engine, style, d_s, read: Callable[[int], Tuple[np.ndarray, Optional[np.ndarray]]], n_frames: int,
                  sink: Callable[[int, np.ndarray], None], batch_size: int = 4, bgr: bool = True,
                  rank: Optional[int] = None, world_size: Optional[int] = None, **kw) -> Tuple[int, int]:
    """This rank's contiguous shard of an `n_frames` video (frames.shard_range): read(i) ->
    (frame, parsing) for absolute frame index i; sink(i, out_frame).  Returns the shard [a, b)."""
    r, _, ws = _frames.world()
    rank = r if rank is None else rank
    world_size = ws if world_size is None else world_size
    a, b = _frames.shard_range(n_frames, rank, world_size)
    vt = VideoToonifier(engine, style, d_s, batch_size=batch_size, bgr=bgr, **kw)
    vt.run((read(i) for i in range(a, b)), sink, first_index=a)
    return a, b
