"""RAFT optical flow on the MI355X kernels -- the flow network of the reference's flicker-reduction pre-pass
(smooth_parsing_map.py:95-102,154: `RAFT(args)`, `raft_model(image1, image2, iters=20, test_mode=True)`;
model/raft/core/raft.py, extractor.py, update.py, corr.py).

`RAFT(args)` keeps the reference's constructor, state_dict schema (179 entries: fnet / cnet / update_block, BatchNorm
buffers included, the shared `norm3` / `downsample.1` entries too) and `forward(image1, image2, iters=12,
flow_init=None, upsample=True, test_mode=False)`; `RaftEngine` runs it:

* every convolution is vt_conv2d (BatchNorm of the context encoder folded into the weights at load time; the 7x7
  stride-2 stems and the (1,5)/(5,1) convs of the SepConvGRU -- `pad_w` of vt_conv_desc -- on the register-staged
  kernel, the 3x3 / 1x1 ones on the tile kernels, the 2-channel flow head on the thin-output kernel);
* torch.cat never happens: producers write channel slices of one 400-channel pixel row
  [net 128 | inp 128 | motion 126 | 0 0 | flow 2 | 0 x14] (the weights are packed to that order), the q-conv reads
  cat[r*h, x] as two sources;
* InstanceNorm of the feature encoder = vt_instnorm_stats + vt_affine_apply, ReLU = fused_bias_act; sigmoid / tanh are
  conv epilogues; r*h, the GRU blend, coords = grid + flow, the convex up-sampling are the small kernels of
  csrc/flow_ops.hip; the correlation lookup is vt_corr_lookup on the 2x2-average pyramid of fmap2 (the memory-
  efficient form: no all-pairs volume; equal to CorrBlock up to summation order);
* fp32 by default (exact-fp32 MFMA): the refinement loop feeds its own output back 12-20 times.

No CPU path: GPU tensors (or the host emulation in tests).  Eager launches -- this runs once per window of a video,
not per frame.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, Optional

import torch
from torch import nn

from . import _lib
from . import kernels as K
from . import raft_corr
from ._lib import ACT_LRELU, ACT_NONE, ACT_SIGMOID, ACT_TANH, OUT_NCHW

BN_EPS = 1e-5
HX = 400          # channels of the GRU input row: [net 128 | inp 128 | motion 126 | 0 0 | flow 2 | 0 x14]
FLOW_AT = 384     # 8-aligned slot of the two flow channels


def _p(t):
    return C.c_void_p(t.data_ptr() if isinstance(t, torch.Tensor) else int(t))


# ---------------------------------------------------------------------------------------------------------
# state_dict schema (model/raft/core/extractor.py:6-60,115-163; update.py:6-139)
# ---------------------------------------------------------------------------------------------------------
def _conv_keys(p, cin, cout, kh, kw):
    return {p + ".weight": (cout, cin, kh, kw), p + ".bias": (cout,)}


def _bn_keys(p, c):
    return {p + ".weight": (c,), p + ".bias": (c,), p + ".running_mean": (c,), p + ".running_var": (c,),
            p + ".num_batches_tracked": ()}


def _encoder_keys(p, out_dim, batch_norm):
    k = {}
    if batch_norm:
        k.update(_bn_keys(p + "norm1", 64))
    k.update(_conv_keys(p + "conv1", 3, 64, 7, 7))
    cin = 64
    for li, (dim, stride) in enumerate(((64, 1), (96, 2), (128, 2)), start=1):
        for bi, (ci, s) in enumerate(((cin, stride), (dim, 1))):
            q = f"{p}layer{li}.{bi}."
            k.update(_conv_keys(q + "conv1", ci, dim, 3, 3))
            k.update(_conv_keys(q + "conv2", dim, dim, 3, 3))
            if batch_norm:
                k.update(_bn_keys(q + "norm1", dim))
                k.update(_bn_keys(q + "norm2", dim))
            if s != 1:
                if batch_norm:
                    k.update(_bn_keys(q + "norm3", dim))
                k.update(_conv_keys(q + "downsample.0", ci, dim, 1, 1))
                if batch_norm:
                    k.update(_bn_keys(q + "downsample.1", dim))     # the same module as norm3, registered twice
        cin = dim
    k.update(_conv_keys(p + "conv2", 128, out_dim, 1, 1))
    return k


def raft_schema() -> Dict[str, tuple]:
    k = {}
    k.update(_encoder_keys("fnet.", 256, False))
    k.update(_encoder_keys("cnet.", 256, True))
    u = "update_block."
    k.update(_conv_keys(u + "encoder.convc1", 324, 256, 1, 1))
    k.update(_conv_keys(u + "encoder.convc2", 256, 192, 3, 3))
    k.update(_conv_keys(u + "encoder.convf1", 2, 128, 7, 7))
    k.update(_conv_keys(u + "encoder.convf2", 128, 64, 3, 3))
    k.update(_conv_keys(u + "encoder.conv", 256, 126, 3, 3))
    for g in ("z", "r", "q"):
        k.update(_conv_keys(u + f"gru.conv{g}1", 384, 128, 1, 5))
    for g in ("z", "r", "q"):
        k.update(_conv_keys(u + f"gru.conv{g}2", 384, 128, 5, 1))
    k.update(_conv_keys(u + "flow_head.conv1", 128, 256, 3, 3))
    k.update(_conv_keys(u + "flow_head.conv2", 256, 2, 3, 3))
    k.update(_conv_keys(u + "mask.0", 128, 256, 3, 3))
    k.update(_conv_keys(u + "mask.2", 256, 576, 1, 1))
    return k


class _Node(nn.Module):
    """A container that only holds parameters / buffers under the reference's names."""


def _register(root: nn.Module, key: str, shape):
    parts = key.split(".")
    m = root
    for name in parts[:-1]:
        if not hasattr(m, name):
            m.add_module(name, _Node())
        m = getattr(m, name)
    leaf = parts[-1]
    if leaf in ("running_mean", "running_var"):
        m.register_buffer(leaf, torch.zeros(shape) if leaf == "running_mean" else torch.ones(shape))
    elif leaf == "num_batches_tracked":
        m.register_buffer(leaf, torch.zeros(shape, dtype=torch.long))
    else:
        m.register_parameter(leaf, nn.Parameter(torch.zeros(shape), requires_grad=False))


class RAFT(nn.Module):
    """model.raft.core.raft.RAFT(args) -- args.small / mixed_precision / alternate_corr are accepted; only the
    configuration the reference builds (not small) exists here."""

    def __init__(self, args=None, compute_dtype=torch.float32):
        super().__init__()
        if args is not None and getattr(args, "small", False):
            raise _lib.VtError("RAFT: the small model is not built (smooth_parsing_map.py uses the full one)")
        self.args = args
        self.hidden_dim, self.context_dim = 128, 128
        self.compute_dtype = compute_dtype
        for key, shape in raft_schema().items():
            _register(self, key, shape)
        self._engine: Optional[RaftEngine] = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate())

    def invalidate(self):
        self._engine = None

    def _apply(self, fn, *a, **k):
        self._engine = None
        return super()._apply(fn, *a, **k)

    def engine(self) -> "RaftEngine":
        if self._engine is None:
            dev = next(self.parameters()).device
            self._engine = RaftEngine({k: v.detach() for k, v in self.state_dict().items()}, self.compute_dtype, dev)
        return self._engine

    def forward(self, image1, image2, iters=12, flow_init=None, upsample=True, test_mode=False):
        flow_low, ups = self.engine().forward(image1, image2, iters=iters, flow_init=flow_init,
                                              all_predictions=not test_mode)
        if test_mode:
            return flow_low, ups[-1]
        return ups


# ---------------------------------------------------------------------------------------------------------
# engine
# ---------------------------------------------------------------------------------------------------------
class RaftEngine:
    def __init__(self, sd: Dict[str, torch.Tensor], dtype=torch.float32, device="cuda"):
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        if self.device.type != "cuda" and not _lib.is_emulation():
            raise _lib.VtError("RaftEngine needs a GPU device (no CPU path)")
        self.dtype, self.dt = dtype, K.dt_code(dtype)
        self.esz = 4 if dtype == torch.float32 else 2
        f32 = lambda k: sd[k].to(self.device, torch.float32)
        self.w: Dict[str, torch.Tensor] = {}
        self.b: Dict[str, torch.Tensor] = {}
        self.keep_taps = False      # tests: keep intermediate tensors of the first iteration in self.taps
        self._graphs = {}           # (shape, iters) -> "seen" | (hipGraph, static inputs, outputs)

        def add(name, key, bn=None, cin_dst=None, chan_map=None, rows=None):
            w, b = f32(key + ".weight"), f32(key + ".bias")
            if bn is not None:   # eval BatchNorm folded into the conv in front of it
                s = f32(bn + ".weight") / torch.sqrt(f32(bn + ".running_var") + BN_EPS)
                w = w * s.view(-1, 1, 1, 1)
                b = (b - f32(bn + ".running_mean")) * s + f32(bn + ".bias")
            if rows is not None:
                w, b = w[rows[0]:rows[1]], b[rows[0]:rows[1]]
            cm = None if chan_map is None else torch.tensor(chan_map, dtype=torch.int32, device=self.device)
            self.w[name] = K.pack_conv_weight(w.contiguous(), cin_dst=cin_dst, chan_map=cm, out_dtype=dtype)
            self.b[name] = b.contiguous()

        for enc, batch in (("fnet.", False), ("cnet.", True)):
            add(enc + "conv1", enc + "conv1", enc + "norm1" if batch else None, cin_dst=8)
            for li, stride in ((1, 1), (2, 2), (3, 2)):
                for bi in (0, 1):
                    q = f"{enc}layer{li}.{bi}."
                    add(q + "conv1", q + "conv1", q + "norm1" if batch else None)
                    add(q + "conv2", q + "conv2", q + "norm2" if batch else None)
                    if stride != 1 and bi == 0:
                        # norm3 and downsample.1 are ONE module registered twice (extractor.py:41-45): load_state_dict
                        # fills it from both entries and the later one, downsample.1, stays
                        add(q + "down", q + "downsample.0", q + "downsample.1" if batch else None)
        add("fnet.conv2", "fnet.conv2")
        add("cnet.net", "cnet.conv2", rows=(0, 128))      # torch.split(cnet, [hdim, cdim]) (raft.py:113)
        add("cnet.inp", "cnet.conv2", rows=(128, 256))
        u = "update_block."
        add("convc1", u + "encoder.convc1", cin_dst=336)
        add("convc2", u + "encoder.convc2")
        add("convf1", u + "encoder.convf1", cin_dst=8)
        add("convf2", u + "encoder.convf2")
        add("conv", u + "encoder.conv")
        # slot of the HX row -> source channel of cat[h, inp, motion features (126 + flow 2)], -1 = zero padding
        cmap = [d if d < 382 else -1 for d in range(HX)]
        cmap[FLOW_AT], cmap[FLOW_AT + 1] = 382, 383
        for g in ("z1", "r1", "q1", "z2", "r2", "q2"):
            add("gru." + g, u + "gru.conv" + g, cin_dst=HX, chan_map=cmap)
        add("fh1", u + "flow_head.conv1")
        add("fh2", u + "flow_head.conv2")
        add("mask0", u + "mask.0")
        add("mask2", u + "mask.2")

    # ------------------------------------------------------------------ helpers
    def _buf(self, shape, dtype=None, zero=False):
        f = torch.zeros if zero else torch.empty
        return f(shape, dtype=dtype or self.dtype, device=self.device)

    def _conv(self, name, src0, c0, ld0, n, h, w, cout, kh, kw, out, ld_out, stride=1, pad=0, pad_w=None, act=ACT_NONE,
              slope=0.0, gain=1.0, src1=None, c1=0, ld1=0, resid=None, ld_res=0, beta=0.0, post_relu=0, planar=False):
        pw = pad if pad_w is None else pad_w
        oh = (h + 2 * pad - (kh - 1) - 1) // stride + 1
        ow = (w + 2 * pw - (kw - 1) - 1) // stride + 1
        kw_ = dict(src0=src0, c0=c0, ld0=ld0, src1=src1, c1=c1, ld1=ld1, n=n, h=h, w=w, out_h=oh, out_w=ow,
                   weight=self.w[name], cout=cout, kh=kh, kw=kw, stride=stride, pad=pad, pad_w=pad_w, bias=self.b[name],
                   act=act, slope=slope, gain=gain, resid=resid, ld_res=ld_res, beta=beta, post_relu=post_relu, out=out,
                   ld_out=ld_out, dtype=self.dt, stream_of=self._anchor)
        if planar:
            kw_.update(out_layout=OUT_NCHW, out_dtype=K.VT_F32, ld_out=0)
        K.conv2d(**kw_)
        return oh, ow

    def _relu_(self, x):
        y = K.fused_bias_act(x, None, None, 3, 0, 0.0, 1.0)      # leaky_relu with slope 0, scale 1
        return y

    def _in_relu(self, x, n, hw, c):
        """relu(InstanceNorm2d(x)) (affine-free, eps 1e-5, biased variance): statistics, affine pass, ReLU."""
        sc, sh = self._buf((n, c), torch.float32), self._buf((n, c), torch.float32)
        ws = self._buf((max(K.instnorm_ws_bytes(n, hw, c), 16),), torch.uint8)
        K.instnorm_stats(sc, sh, x, c, n, hw, c, ws, self.dt, stream_of=x)
        y = torch.empty_like(x)
        K.affine_apply(y, c, x, c, sc, sh, n, hw, c, self.dt, stream_of=x)
        return self._relu_(y)

    def _encoder(self, enc, x8, n, H, W, instance):
        """BasicEncoder up to layer3 (extractor.py:165-183): x8 (n,H,W,8) NHWC, 3 real channels -> (n,H/8,W/8,128)."""
        relu = dict(act=ACT_LRELU, slope=0.0)
        h, w = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        x = self._buf((n, h, w, 64))
        self._conv(enc + "conv1", x8, 8, 8, n, H, W, 64, 7, 7, x, 64, stride=2, pad=3, **({} if instance else relu))
        if instance:
            x = self._in_relu(x, n, h * w, 64)
        cin = 64
        for li, (dim, stride) in enumerate(((64, 1), (96, 2), (128, 2)), start=1):
            for bi, (ci, s) in enumerate(((cin, stride), (dim, 1))):
                q = f"{enc}layer{li}.{bi}."
                oh, ow = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
                y1 = self._buf((n, oh, ow, dim))
                self._conv(q + "conv1", x, ci, ci, n, h, w, dim, 3, 3, y1, dim, stride=s, pad=1, **({} if instance else relu))
                if instance:
                    y1 = self._in_relu(y1, n, oh * ow, dim)
                short = x
                if s != 1:
                    short = self._buf((n, oh, ow, dim))
                    self._conv(q + "down", x, ci, ci, n, h, w, dim, 1, 1, short, dim, stride=s)
                    if instance:     # norm3 is an InstanceNorm without ReLU
                        sc, sh = self._buf((n, dim), torch.float32), self._buf((n, dim), torch.float32)
                        ws = self._buf((max(K.instnorm_ws_bytes(n, oh * ow, dim), 16),), torch.uint8)
                        K.instnorm_stats(sc, sh, short, dim, n, oh * ow, dim, ws, self.dt, stream_of=short)
                        s2 = torch.empty_like(short)
                        K.affine_apply(s2, dim, short, dim, sc, sh, n, oh * ow, dim, self.dt, stream_of=short)
                        short = s2
                y2 = self._buf((n, oh, ow, dim))
                if instance:
                    self._conv(q + "conv2", y1, dim, dim, n, oh, ow, dim, 3, 3, y2, dim, pad=1)
                    y2 = self._in_relu(y2, n, oh * ow, dim)
                    out = self._buf((n, oh, ow, dim))
                    _lib.check(self.lib.vt_eltwise2(_p(out), dim, _p(short), dim, _p(y2), dim, n * oh * ow, dim, 2,
                                                    self.dt, self._st()), "vt_eltwise2")
                else:   # relu(x + relu(bn2(conv2(y)))) in one launch: act, residual add, post_relu
                    out = y2
                    self._conv(q + "conv2", y1, dim, dim, n, oh, ow, dim, 3, 3, out, dim, pad=1, resid=short, ld_res=dim,
                               beta=1.0, post_relu=1, **relu)
                x, h, w = out, oh, ow
            cin = dim
        return x, h, w

    def _st(self):
        return K._stream(self._anchor)

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, image1: torch.Tensor, image2: torch.Tensor, iters: int = 12, flow_init=None,
                all_predictions: bool = False):
        """image1, image2 (N,3,H,W) in [0,255], H and W multiples of 8 (InputPadder of the caller).  Returns
        (flow_low (N,2,H/8,W/8), [flow_up (N,2,H,W)] -- every iteration's when all_predictions, else the last).

        VT_RAFT_GRAPH=1: the inference form (test_mode: no flow_init, last prediction only) as ONE hipGraph replay per
        (N, H, W, iters), captured on the second call of a shape; bit-identical to the eager launches.  OFF by default:
        measured (round 3, profiles/r03_raft_bench.txt) it changes nothing -- 116.5 vs 117.1 ms for 11 pairs of 512x512 x 20
        iterations, 29.0 vs 29.2 ms for one pair: the ~1 300 launches are GPU-bound (fp32 convolutions on 64x64-pixel
        maps: tiny grids), not host-bound.  What does move it is the arithmetic: RAFT(..., compute_dtype=torch.bfloat16)
        runs the same window in 62.6 ms (5.7 ms per pair); its deviation from the fp32 flow cannot be judged on synthetic
        weights (they produce 600-pixel flows), so fp32 stays the default."""
        if (self.device.type == "cuda" and flow_init is None and not all_predictions and not self.keep_taps and
                os.environ.get("VT_RAFT_GRAPH", "0") == "1" and image1.shape == image2.shape and image1.ndim == 4):
            key = (tuple(image1.shape), int(iters))
            ent = self._graphs.get(key)
            if ent is None:
                self._graphs[key] = "seen"          # first call of a shape: eager (also the warm-up of the capture)
            else:
                if ent == "seen":
                    if len(self._graphs) > 8:
                        self._graphs.clear()
                    i1 = torch.empty(image1.shape, dtype=torch.float32, device=self.device)
                    i2 = torch.empty_like(i1)
                    i1.copy_(image1)
                    i2.copy_(image2)
                    torch.cuda.synchronize(self.device)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        lo, ups = self._forward(i1, i2, iters, None, False)
                    ent = self._graphs[key] = (g, i1, i2, lo, ups[-1])
                g, i1, i2, lo, up = ent
                i1.copy_(image1)
                i2.copy_(image2)
                g.replay()
                return lo.clone(), [up.clone()]
        return self._forward(image1, image2, iters, flow_init, all_predictions)

    def _forward(self, image1, image2, iters, flow_init, all_predictions):
        self.lib = _lib.lib()
        if image1.shape != image2.shape or image1.ndim != 4 or image1.shape[1] != 3:
            raise _lib.VtError("RAFT: image1 / image2 must be (N,3,H,W) of the same shape")
        N, _, H, W = image1.shape
        if H % 8 or W % 8:
            raise _lib.VtError("RAFT: H and W must be multiples of 8 (pad with InputPadder first)")
        image1 = image1.to(self.device, torch.float32).contiguous()
        image2 = image2.to(self.device, torch.float32).contiguous()
        self._anchor = image1
        K._dev_ok(image1, image2)
        # 2 * (x / 255) - 1 (raft.py:89-90) on the NHWC copy; the 5 padding channels stay zero
        sc = torch.zeros((2 * N, 8), device=self.device)
        sh = torch.zeros((2 * N, 8), device=self.device)
        sc[:, :3], sh[:, :3] = 2.0 / 255.0, -1.0
        both = K.nchw_to_nhwc(torch.cat([image1, image2], 0), self.dtype)     # (2N,H,W,8): fnet([image1, image2])
        xn = torch.empty_like(both)
        K.affine_apply(xn, 8, both, 8, sc, sh, 2 * N, H * W, 8, self.dt, stream_of=both)

        # feature network (instance norm) on both frames, context network (folded batch norm) on the first
        f, h, w = self._encoder("fnet.", xn, 2 * N, H, W, True)
        fmap = self._buf((2 * N, h, w, 256), torch.float32)
        self._conv("fnet.conv2", f, 128, 128, 2 * N, h, w, 256, 1, 1, self._fmap_out(fmap), 256)
        fmap = self._fmap_done(fmap)
        fmap1, fmap2 = fmap[:N].contiguous(), fmap[N:].contiguous()
        pyramid = [fmap2]
        for _ in range(3):     # corr_levels = 4
            pyramid.append(raft_corr.avg_pool2x2_nhwc(pyramid[-1]))
        c, h, w = self._encoder("cnet.", xn[:N].contiguous(), N, H, W, False)
        hx = self._buf((N, h, w, HX), zero=True)
        base = hx.data_ptr()
        self._conv("cnet.net", c, 128, 128, N, h, w, 128, 1, 1, base, HX, act=ACT_TANH)                       # net = tanh(.)
        self._conv("cnet.inp", c, 128, 128, N, h, w, 128, 1, 1, base + 128 * self.esz, HX, act=ACT_LRELU, slope=0.0)

        taps = {"cnet": hx[..., :256].clone()} if self.keep_taps else {}
        flow = torch.zeros((N, 2, h, w), device=self.device)              # coords1 - coords0
        if flow_init is not None:
            flow += flow_init.to(self.device, torch.float32)
        coords = self._buf((N, 1, h, w, 2), torch.float32)
        corr_in = self._buf((N, h, w, 336), zero=True)
        cor_flo = self._buf((N, h, w, 256))
        flow8 = self._buf((N, h, w, 8), zero=True)
        z, r, rh, q = (self._buf((N, h, w, 128)) for _ in range(4))
        relu = dict(act=ACT_LRELU, slope=0.0)
        rows = N * h * w
        scale = 1.0 / math.sqrt(256.0)
        ups = []
        for itr in range(iters):
            _lib.check(self.lib.vt_coords_from_flow(_p(coords), _p(flow), N, h, w, self._st()), "vt_coords_from_flow")
            corr = torch.stack([raft_corr.forward(fmap1, pyramid[i], coords, 4, scale=scale, coord_scale=1.0 / 2 ** i)[0]
                                .squeeze(1) for i in range(4)], dim=1).reshape(N, 324, h, w)   # corr.py:88-91
            if self.keep_taps and itr == 0:
                taps["corr1"] = corr.clone()
            K.nchw_to_nhwc(corr, self.dtype, ld_out=336, out=corr_in)
            # BasicMotionEncoder (update.py:93-108)
            t256 = self._buf((N, h, w, 256))
            self._conv("convc1", corr_in, 336, 336, N, h, w, 256, 1, 1, t256, 256, **relu)
            self._conv("convc2", t256, 256, 256, N, h, w, 192, 3, 3, cor_flo, 256, pad=1, **relu)
            K.nchw_to_nhwc(flow, self.dtype, ld_out=8, out=flow8)
            t128 = self._buf((N, h, w, 128))
            self._conv("convf1", flow8, 8, 8, N, h, w, 128, 7, 7, t128, 128, pad=3, **relu)
            self._conv("convf2", t128, 128, 128, N, h, w, 64, 3, 3, cor_flo.data_ptr() + 192 * self.esz, 256, pad=1, **relu)
            self._conv("conv", cor_flo, 256, 256, N, h, w, 126, 3, 3, base + 256 * self.esz, HX, pad=1, **relu)
            _lib.check(self.lib.vt_nchw_to_nhwc(C.c_void_p(base + FLOW_AT * self.esz), HX, _p(flow), N, 2, h * w, K.VT_F32,
                                                self.dt, self._st()), "vt_nchw_to_nhwc")
            # SepConvGRU (update.py:33-57): horizontal (1,5) then vertical (5,1)
            for tag, kh, kw, pad, pad_w in (("1", 1, 5, 0, 2), ("2", 5, 1, 2, 0)):
                self._conv("gru.z" + tag, hx, HX, HX, N, h, w, 128, kh, kw, z, 128, pad=pad, pad_w=pad_w, act=ACT_SIGMOID)
                self._conv("gru.r" + tag, hx, HX, HX, N, h, w, 128, kh, kw, r, 128, pad=pad, pad_w=pad_w, act=ACT_SIGMOID)
                _lib.check(self.lib.vt_eltwise2(_p(rh), 128, _p(r), 128, _p(hx), HX, rows, 128, 0, self.dt, self._st()),
                           "vt_eltwise2")
                self._conv("gru.q" + tag, rh, 128, 128, N, h, w, 128, kh, kw, q, 128, pad=pad, pad_w=pad_w, act=ACT_TANH,
                           src1=base + 128 * self.esz, c1=HX - 128, ld1=HX)
                _lib.check(self.lib.vt_gru_blend(_p(hx), HX, _p(z), _p(q), rows, 128, self.dt, self._st()), "vt_gru_blend")
            # FlowHead (update.py:6-14); coords1 = coords1 + delta_flow (raft.py:127)
            self._conv("fh1", hx, 128, HX, N, h, w, 256, 3, 3, t256, 256, pad=1, **relu)
            if self.keep_taps and itr == 0:
                taps["net1"], taps["flow0"] = hx[..., :128].clone(), flow.clone()
            self._conv("fh2", t256, 256, 256, N, h, w, 2, 3, 3, flow, 0, pad=1, resid=flow, beta=1.0, planar=True)
            if self.keep_taps and itr == 0:
                taps["delta1"] = flow - taps["flow0"]
            if all_predictions or itr == iters - 1:
                m256 = self._buf((N, h, w, 256))
                self._conv("mask0", hx, 128, HX, N, h, w, 256, 3, 3, m256, 256, pad=1, **relu)
                mask = self._buf((N, 576, h, w), torch.float32)
                self._conv("mask2", m256, 256, 256, N, h, w, 576, 1, 1, mask, 0, gain=0.25, planar=True)   # .25 * mask(net)
                up = self._buf((N, 2, 8 * h, 8 * w), torch.float32)
                _lib.check(self.lib.vt_convex_upsample(_p(up), _p(flow), _p(mask), N, h, w, self._st()),
                           "vt_convex_upsample")
                ups.append(up)
        taps["fmap1"] = fmap1
        self.taps = taps
        return flow.clone(), ups

    # fmap is consumed by the fp32 correlation lookup: in fp32 mode the conv writes it directly
    def _fmap_out(self, fmap):
        if self.dtype == torch.float32:
            return fmap
        self._fmap_lo = torch.empty(fmap.shape, dtype=self.dtype, device=self.device)
        return self._fmap_lo

    def _fmap_done(self, fmap):
        return fmap if self.dtype == torch.float32 else self._fmap_lo.float()
