"""Operator parity: HIP kernels vs the CPU oracle and the reference's golden vectors.

Every test runs on two backends (conftest.dev): the host emulation of the kernel sources
(CPU, not gpu) and the product library on a real MI355X (-m gpu).  Tolerances:
  * upfirdn2d on integer-valued inputs / dyadic taps: BIT-EXACT
  * fused_leaky_relu fp32: BIT-EXACT (same add / select-mul / mul roundings)
  * fp32 contractions: 2e-5 of the output max-abs (fp32 MFMA = fmaf chain, other order)
  * bf16 contractions: 2e-2 of the output max-abs (inputs rounded to bf16, fp32 accumulate)
"""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import vtoonify_oracle as O
from vtoonify_amd import kernels as K
from vtoonify_amd import op

F32_TOL, BF16_TOL = 2e-5, 2e-2


def T(a, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t.to(dtype) if dtype is not None else t


def tol(dtype):
    return F32_TOL if dtype == torch.float32 else BF16_TOL


# ------------------------------------------------------------------------------ upfirdn2d
def test_upfirdn2d_golden(dev):
    d, meta = load_golden("op_upfirdn2d.npz")
    for m in meta:
        n = m["name"]
        up = tuple(m["up"]) if isinstance(m["up"], list) else m["up"]
        down = tuple(m["down"]) if isinstance(m["down"], list) else m["down"]
        y = op.upfirdn2d(T(d[n + "__x"], dev), T(d[n + "__k"], dev), up=up, down=down, pad=tuple(m["pad"]))
        ref = d[n + "__y"]
        assert tuple(y.shape) == ref.shape, n
        if m["integer"]:
            assert np.array_equal(y.cpu().numpy(), ref), f"{n}: index arithmetic not bit-exact"
        else:
            assert rel_err(y.cpu().numpy(), ref) < 1e-6, n


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_upfirdn2d_dtypes_and_edges(dev, dtype):
    g = np.random.default_rng(5)
    k = (np.outer([1, 3, 3, 1], [1, 3, 3, 1]) / 64.0).astype(np.float32)
    # sizes straddling the 32x64 tile, odd sizes 2h+1 (blur after conv_transpose)
    for shape, up, down, pad in [((3, 2, 65, 129), 1, 1, (1, 1)), ((1, 3, 33, 31), 2, 1, (2, 1)),
                                 ((2, 2, 67, 130), 1, 2, (1, 1)), ((1, 1, 5, 300), 1, 1, (2, 1)),
                                 ((1, 2, 9, 7), (1, 2), (2, 1), (1, 0, 2, 3))]:
        x = g.integers(-8, 9, shape).astype(np.float32)
        y = op.upfirdn2d(T(x, dev, dtype), T(k * 4, dev), up=up, down=down, pad=pad)
        assert y.dtype == dtype
        ref = O.upfirdn2d(x, k * 4, up, down, pad)
        if dtype == torch.float32:
            assert np.array_equal(y.cpu().numpy(), ref)
        else:  # result rounded once to the 16-bit type
            assert rel_err(y.float().cpu().numpy(), ref) < (8e-3 if dtype == torch.bfloat16 else 1e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_upfirdn2d_wide_plane_kernel(dev, dtype):
    """Planes >= 192 output columns with up == down == 1 take the 16x256-tile kernel (4x4 outputs per
    lane): tile edges, odd widths (unaligned rows -> scalar stores), crops, non-square FIRs."""
    g = np.random.default_rng(6)
    k4 = (np.outer([1, 3, 3, 1], [1, 3, 3, 1]) / 16.0).astype(np.float32)
    k32 = (np.array([[1., 2], [3, 4], [5, 6]]) / 16.0).astype(np.float32)   # asymmetric: catches a missing flip
    for shape, k, pad in [((2, 3, 35, 513), k4, (1, 1)), ((1, 2, 18, 260), k4, (1, 1)), ((1, 1, 16, 259), k4, (2, 2)),
                          ((1, 2, 21, 300), k32, (1, 0, 0, 2)), ((1, 1, 40, 270), k4, (-1, 3, 2, -2)),
                          ((1, 2, 17, 1025), k4, (1, 1))]:
        x = g.integers(-8, 9, shape).astype(np.float32)
        y = op.upfirdn2d(T(x, dev, dtype), T(k, dev), pad=pad)
        ref = O.upfirdn2d(x, k, 1, 1, pad)
        assert y.shape == ref.shape
        if dtype == torch.float32:
            assert np.array_equal(y.cpu().numpy(), ref), (shape, pad)
        else:
            assert rel_err(y.float().cpu().numpy(), ref) < 8e-3, (shape, pad)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_upfirdn2d_wide_plane_kernel_16bit_phases(dev, dtype):
    """The wide tile on 16-bit tensors at every 2-byte phase (written in round 6 for a loader that kept the window raw in LDS --
    measured slower and dropped, profiles/r06_upfirdn2d_experiments.txt; the cases stay as a regression for any loader): plane
    widths of all residues mod 8 (consecutive rows walk through all phases), tensors that do not start on a 16-byte boundary
    (views at element offsets 1..7), first and last plane with a foreign neighbour on both sides, left / right / top / bottom
    edge tiles with pads and crops, a 1 x 4 and a 4 x 1 FIR.  Integer data, dyadic taps: the result is the exact sum rounded
    ONCE to the 16-bit type."""
    g = np.random.default_rng(16)
    k4 = (np.outer([1, 3, 3, 1], [1, 3, 3, 1]) / 16.0).astype(np.float32)
    row = (np.array([[1.0, 3, 3, 1]]) / 4.0).astype(np.float32)
    for w in (513, 514, 515, 516, 517, 518, 519, 520):
        for off, k, pad in ((w % 8, k4, (1, 1)), ((w + 3) % 8, row, (2, 1, 0, 0)), (0, row.T.copy(), (0, 0, 1, 2)), (5, k4, (-2, 3, 1, -1))):
            shape = (2, 2, 18, w)
            x = g.integers(-8, 9, shape).astype(np.float32)
            n = int(np.prod(shape))
            buf = torch.full((n + 16,), 77.0, dtype=dtype, device=dev)      # a neighbour that must never leak into the result
            xt = buf[off:off + n].view(shape)
            xt.copy_(T(x, dev, dtype))
            y = op.upfirdn2d(xt, T(k, dev), pad=pad)
            ref = O.upfirdn2d(x, k, 1, 1, pad)
            assert y.shape == ref.shape and y.dtype == dtype
            # small integers, dyadic taps: the fp32 sums are exact, so the result is `ref` rounded ONCE to the 16-bit type
            want = torch.from_numpy(ref).to(dtype).float().numpy()
            assert np.array_equal(y.float().cpu().numpy(), want), (w, off, pad)


def test_upfirdn2d_properties(dev):
    """Size-independent checks at a realistic size: linearity and the identity kernel."""
    g = torch.Generator().manual_seed(3)
    a = torch.randn(1, 4, 257, 255, generator=g).to(dev)
    b = torch.randn(1, 4, 257, 255, generator=g).to(dev)
    k = torch.tensor(np.outer([1, 3, 3, 1], [1, 3, 3, 1]) / 16.0, dtype=torch.float32)
    f = lambda t: op.upfirdn2d(t, k, pad=(1, 1))
    lhs, rhs = f(a + 2 * b), f(a) + 2 * f(b)
    assert (lhs - rhs).abs().max().item() < 1e-4
    ident = torch.ones(1, 1)
    assert torch.equal(op.upfirdn2d(a, ident), a)
    z = op.upfirdn2d(a, ident, up=2)  # pure zero insertion
    assert torch.equal(z[:, :, ::2, ::2], a) and float(z[:, :, 1::2].abs().max()) == 0.0


def test_upfirdn2d_errors(dev):
    x = torch.zeros(1, 1, 4, 4, device=dev)
    k = torch.ones(4, 4)
    with pytest.raises(Exception, match="empty output"):
        op.upfirdn2d(x, k, pad=(0, 0, -2, -2))
    with pytest.raises(ValueError):
        op.upfirdn2d(x[0], k)


def test_upfirdn2d_backward_matches_oracle_adjoint(dev):
    g = np.random.default_rng(9)
    x = g.standard_normal((1, 2, 9, 8)).astype(np.float32)
    k = (np.outer([1, 3, 3, 1], [1, 2, 2, 1]) / 48.0).astype(np.float32)
    xt = T(x, dev).requires_grad_(True)
    y = op.upfirdn2d(xt, T(k, dev), up=2, down=1, pad=(2, 1))
    gy = g.standard_normal(tuple(y.shape)).astype(np.float32)
    y.backward(T(gy, dev))
    # adjoint identity <A x, gy> = <x, A^T gy> checked against the oracle's forward
    lhs = float((O.upfirdn2d(x, k, 2, 1, (2, 1)) * gy).sum())
    rhs = float((x * xt.grad.cpu().numpy()).sum())
    assert abs(lhs - rhs) < 1e-3 * max(1.0, abs(lhs))


# ------------------------------------------------------------------------ fused_leaky_relu
def test_fused_leaky_relu_golden_bit_exact(dev):
    d, meta = load_golden("op_fused_act.npz")
    for m in meta:
        n = m["name"]
        b = T(d[n + "__b"], dev) if m["has_bias"] else None
        y = op.fused_leaky_relu(T(d[n + "__x"], dev), b, m["slope"], m["scale"])
        assert np.array_equal(y.cpu().numpy(), d[n + "__y"]), n


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fused_leaky_relu_half_types(dev, dtype):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 6, 33, 17, generator=g)
    b = torch.randn(6, generator=g)
    y = op.fused_leaky_relu(x.to(dev, dtype), b.to(dev, dtype))
    ref = O.fused_leaky_relu(x.to(dtype).float().numpy(), b.to(dtype).float().numpy())
    assert y.dtype == dtype and rel_err(y.float().cpu().numpy(), ref) < (8e-3 if dtype == torch.bfloat16 else 1e-3)


def test_fused_leaky_relu_module_and_grad(dev):
    m = op.FusedLeakyReLU(5).to(dev)
    with torch.no_grad():
        m.bias.copy_(torch.linspace(-1, 1, 5))
    x = torch.randn(3, 5, 8, 8, generator=torch.Generator().manual_seed(2)).to(dev).requires_grad_(True)
    y = m(x)
    ref = O.fused_leaky_relu(x.detach().cpu().numpy(), m.bias.detach().cpu().numpy())
    assert np.array_equal(y.detach().cpu().numpy(), ref)
    y.sum().backward()
    xb = x.detach().cpu().numpy() + m.bias.detach().cpu().numpy().reshape(1, 5, 1, 1)
    gref = np.where(xb > 0, 1.0, 0.2).astype(np.float32) * np.float32(2 ** 0.5)
    assert np.allclose(x.grad.cpu().numpy(), gref, rtol=0, atol=1e-6)
    assert np.allclose(m.bias.grad.cpu().numpy(), gref.sum((0, 2, 3)), rtol=1e-5)


# ------------------------------------------------------------------------ MFMA lane maps
@pytest.mark.parametrize("odt", [torch.bfloat16, torch.float32])
def test_layout_change_at_the_model_boundary(dev, odt):
    """vt_nchw_to_nhwc: exact (a cast and a transpose), pad channels zero, rows wider than the padded channel count untouched.
    Narrow tensors (the 22-channel frame) take the one-thread-per-pixel kernel, wide ones the (pixel, 8-channel group) form."""
    g = np.random.default_rng(3)
    for n, c, h, w, ld, idt in ((2, 22, 9, 13, 32, None), (1, 3, 5, 7, None, None), (2, 32, 4, 6, None, None), (1, 40, 3, 5, 48, None),
                                (3, 19, 8, 8, 24, None),
                                # wide tensors: 64-pixel x 64-channel tiles through LDS (ragged pixel and channel tails, several tiles)
                                (2, 128, 9, 13, None, None), (1, 324, 5, 7, 336, None), (2, 70, 8, 9, 72, torch.bfloat16),
                                (1, 64, 70, 3, None, torch.bfloat16), (2, 130, 11, 12, 144, None)):
        x = T(g.standard_normal((n, c, h, w)).astype(np.float32), dev)
        if idt is not None:
            x = x.to(idt)
        cpad = (c + 7) // 8 * 8
        ldo = ld or cpad
        out = torch.full((n, h, w, ldo), 7.0, dtype=odt, device=dev)
        K.nchw_to_nhwc(x, odt, ld_out=ldo, out=out)
        want = torch.full((n, h, w, ldo), 7.0, dtype=odt)
        want[..., :cpad] = 0
        want[..., :c] = x.cpu().permute(0, 2, 3, 1).to(odt)
        assert torch.equal(out.cpu(), want), (n, c, h, w, ld)


@pytest.mark.parametrize("odt", [torch.bfloat16, torch.float32])
def test_layout_change_back_to_planes(dev, odt):
    """vt_nhwc_to_nchw: exact (a cast and a transpose) for rows wider than the channel count; channel counts that are a multiple
    of 8 (>= 32) go through the LDS tiles, the others through the one-thread-per-element form."""
    g = np.random.default_rng(4)
    for n, c, h, w, ld, idt in ((2, 19, 8, 8, 24, torch.float32), (1, 3, 5, 7, 8, torch.bfloat16), (2, 128, 9, 13, 128, torch.bfloat16),
                                (1, 64, 70, 3, 72, torch.float32), (2, 72, 8, 9, 80, torch.bfloat16), (1, 136, 11, 12, 136, torch.float32),
                                (1, 32, 4, 6, 32, torch.float32), (1, 40, 6, 5, 44, torch.float32)):   # (ld 44 fp32: rows stay 16-byte aligned)
        x = T(g.standard_normal((n, h, w, ld)).astype(np.float32), dev).to(idt)
        y = K.nhwc_to_nchw(x, ld, n, c, h, w, idt, odt, x.device, x)
        want = x.cpu()[..., :c].permute(0, 3, 1, 2).to(odt)
        assert torch.equal(y.cpu(), want), (n, c, h, w, ld, idt)


def test_mfma_lane_maps(dev):
    """Pins the 16x16x32 bf16 / 16x16x4 f32 fragment layouts used by conv_igemm.hip with an
    ASYMMETRIC B (a transposed C-write would fail)."""
    g = torch.Generator().manual_seed(7)
    a = torch.randn(16, 16, generator=g)
    b = torch.randn(16, 16, generator=g) + torch.arange(16).float()[:, None]
    c = K.mfma_selftest(a.to(dev), b.to(dev)).cpu()
    assert (c - a @ b.T).abs().max().item() < 1e-4
    a2 = torch.randn(16, 32, generator=g).bfloat16()
    b2 = (torch.randn(16, 32, generator=g) + torch.arange(16).float()[:, None]).bfloat16()
    c2 = K.mfma_selftest(a2.to(dev), b2.to(dev)).cpu()
    assert (c2 - a2.float() @ b2.float().T).abs().max().item() < 1e-3


# ----------------------------------------------------------------------------- conv_igemm
def _conv_case(dev, dtype, N, Cin, H, W, Cout, k, stride, pad, dil, act=0, resid=False, planar=False, hint=0,
               seed=0, ws=False, expect_kind=None):
    g = np.random.default_rng(seed)
    x = g.standard_normal((N, Cin, H, W)).astype(np.float32)
    w = (g.standard_normal((Cout, Cin, k, k)) / math.sqrt(Cin * k * k)).astype(np.float32)
    b = g.standard_normal(Cout).astype(np.float32)
    cpad = (Cin + 7) // 8 * 8
    xt = K.nchw_to_nhwc(T(x, dev), dtype)
    wp = K.pack_conv_weight(T(w, dev), cin_dst=cpad, out_dtype=dtype)
    Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    # the oracle sees the same operand rounding as the kernel (bf16 inputs, fp32 accumulate)
    xq = xt.float().cpu().permute(0, 3, 1, 2)[:, :Cin].numpy()
    wq = wp.float().cpu().numpy().reshape(Cout, k, k, cpad)[..., :Cin].transpose(0, 3, 1, 2)
    ref = O.conv2d(xq, wq, b, stride, pad, dil)
    gain = 1.0
    if act == K.ACT_LRELU:
        ref, gain = O.leaky_relu(ref, 0.2) * np.float32(2 ** 0.5), 2 ** 0.5
    elif act == K.ACT_RELU_TANH:
        ref = np.tanh(np.maximum(ref, 0))
    common = dict(src0=xt, c0=cpad, ld0=cpad, n=N, h=H, w=W, out_h=Ho, out_w=Wo, weight=wp, cout=Cout, kh=k, kw=k,
                  stride=stride, pad=pad, dil=dil, bias=T(b, dev), act=act, gain=gain, dtype=K.dt_code(dtype),
                  tile_hint=hint, alpha=0.5 if resid else 1.0, beta=0.25 if resid else 0.0)
    if ws:  # split-K workspace: lets the heuristics (or the hint) cut K across workgroups
        common["splitk_ws"] = torch.zeros(8 << 20, dtype=torch.float32, device=dev)
    if expect_kind is not None:
        import ctypes
        from vtoonify_amd import _lib
        d = K.make_conv_desc(out=xt, ld_out=8, **common)
        code = _lib.lib().vt_conv2d_tile(ctypes.byref(d))
        assert (code // 100000000 == 1) == (expect_kind == 1), f"kernel kind {code}"
    if planar:
        out = torch.zeros((N, Cout, Ho, Wo), dtype=torch.float32, device=dev)
        r = None
        if resid:
            rn = g.standard_normal((N, Cout, Ho, Wo)).astype(np.float32)
            r, ref = T(rn, dev), ref * 0.5 + 0.25 * rn
        K.conv2d(out=out, ld_out=0, resid=r, out_layout=K.OUT_NCHW, out_dtype=K.VT_F32, **common)
        y = out.cpu().numpy()
    else:
        ldo = (Cout + 7) // 8 * 8
        out = torch.zeros((N, Ho, Wo, ldo), dtype=dtype, device=dev)
        r = None
        if resid:
            rn = g.standard_normal((N, Cout, Ho, Wo)).astype(np.float32)
            r = K.nchw_to_nhwc(T(rn, dev), dtype, ld_out=ldo)
            ref = ref * 0.5 + 0.25 * r.float().cpu().permute(0, 3, 1, 2)[:, :Cout].numpy()
        K.conv2d(out=out, ld_out=ldo, resid=r, ld_res=ldo, **common)
        y = out.float().cpu().permute(0, 3, 1, 2)[:, :Cout].numpy()
    # operands were pre-rounded, so only accumulation order and the output rounding remain
    return rel_err(y, ref)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_shapes(dev, dtype):
    t = F32_TOL if dtype == torch.float32 else 8e-3  # bf16: output rounding only
    L, RT = K.ACT_LRELU, K.ACT_RELU_TANH
    assert _conv_case(dev, dtype, 1, 22, 12, 9, 32, 3, 1, 1, 1, act=L) < t        # stem: 22 -> pad 24
    assert _conv_case(dev, dtype, 2, 32, 13, 10, 64, 3, 2, 1, 1, act=L, resid=True) < t   # stride 2, odd size
    assert _conv_case(dev, dtype, 1, 64, 9, 9, 64, 3, 1, 2, 2, act=L) < t         # dilation 2
    assert _conv_case(dev, dtype, 1, 40, 9, 11, 128, 3, 1, 4, 4) < t              # dilation 4, K tail
    assert _conv_case(dev, dtype, 2, 64, 7, 5, 3, 1, 1, 0, 1, planar=True, resid=True) < t   # ToRGB-like
    assert _conv_case(dev, dtype, 1, 72, 8, 8, 1, 3, 1, 1, 1, act=RT, planar=True) < t       # mask conv
    assert _conv_case(dev, dtype, 1, 16, 6, 6, 20, 3, 1, 1, 1) < t                # cout tail (scalar stores)
    assert _conv_case(dev, dtype, 1, 8, 1, 1, 8, 3, 1, 1, 1) < t                  # single pixel


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_thin_kernel(dev, dtype):
    """conv_thin.hpp (KIND 6 of vt_conv2d_tile): cout <= 3, planar output, 3x3 / 1x1 -- the ToRGB, Fusion-mask and
    fusion_skip convs (vtoonify.py:111,126,176,197-198; model.py:383-392) as one launch in the scatter form
    (GEMM over the input pixels with taps*cout virtual channels, then a 9-term stencil).  Tile edges (sizes that
    are not multiples of 8), batches, K-steps that do not divide by the 4 wavefronts, residual, relu+tanh."""
    import ctypes
    from vtoonify_amd import _lib
    t = F32_TOL if dtype == torch.float32 else 8e-3
    RT = K.ACT_RELU_TANH
    kstep = 16 if dtype == torch.float32 else 32
    x0 = torch.zeros((1, 8, 8, 4 * kstep), dtype=dtype, device=dev)
    w0 = torch.zeros((3, 9, 4 * kstep), dtype=dtype, device=dev)
    o0 = torch.zeros((1, 3, 8, 8), dtype=torch.float32, device=dev)
    d = K.make_conv_desc(src0=x0, c0=4 * kstep, ld0=4 * kstep, n=1, h=8, w=8, out_h=8, out_w=8, weight=w0, cout=3, kh=3,
                         kw=3, pad=1, out=o0, ld_out=0, out_layout=K.OUT_NCHW, out_dtype=K.VT_F32, dtype=K.dt_code(dtype))
    assert _lib.lib().vt_conv2d_tile(ctypes.byref(d)) // 100000000 == 6
    for (N, Cin, H, W, Cout, k, act, resid) in [
            (1, 4 * kstep, 8, 8, 1, 3, RT, False),          # mask conv, one tile
            (2, 5 * kstep, 11, 13, 3, 3, 0, True),          # fusion_skip-like: 27 virtual channels, edges, batch 2
            (1, 18 * kstep, 9, 7, 3, 3, 0, False),          # 18 K-steps over 4 waves (576 channels in bf16)
            (1, 2 * kstep, 5, 20, 2, 3, 0, False),          # fewer K-steps than wavefronts; cout 2
            (2, 4 * kstep, 7, 5, 3, 1, 0, True),            # ToRGB: 1x1 + bias + up-sampled skip (residual)
            (1, 9 * kstep, 16, 8, 1, 1, 0, False),
            # >= 256 tiles of 16 x 16 per image: conv_thin16_kernel (round 6; pixels split over the waves, K in one chain)
            (1, 3 * kstep, 256, 256, 1, 3, RT, False),      # mask conv shape, whole tiles, odd number of K-steps
            (2, 2 * kstep, 258, 262, 3, 3, 0, True)]:       # fusion_skip-like: 27 virtual channels, ragged edges, batch 2, residual
        pad = k // 2
        e_new = _conv_case(dev, dtype, N, Cin, H, W, Cout, k, 1, pad, 1, act=act, resid=resid, planar=True, seed=Cin + H)
        e_old = _conv_case(dev, dtype, N, Cin, H, W, Cout, k, 1, pad, 1, act=act, resid=resid, planar=True, seed=Cin + H,
                           hint=128016, ws=True)
        assert e_new < t and e_old < t, (N, Cin, H, W, Cout, k, e_new, e_old)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_thin_gate_prologue(dev, dtype):
    """vt_conv_desc.in_absdiff + in_scale / in_shift on the thin kernel: the Fusion gate's mask conv reads f_G and f_E, forms
    cat[f_G, |f_G - f_E|] and applies the AdaIN affine in its loader (vtoonify.py:125-126) -- BIT-identical to
    vt_affine_apply into a 2C-channel tensor followed by the plain conv; tile edges, batch 2, K-steps that do not divide
    by the wavefronts; refused on the other kernels."""
    import ctypes as C
    from vtoonify_amd import _lib
    lib = _lib.lib()
    g = np.random.default_rng(41)
    kstep = 16 if dtype == torch.float32 else 32
    code = K.dt_code(dtype)
    for (N, c, H, W) in ((1, 2 * kstep, 8, 8), (2, 3 * kstep, 11, 13), (1, 5 * kstep, 5, 20),
                         (2, 2 * kstep, 256, 260), (1, 3 * kstep, 258, 256)):   # (the last two: 16 x 16 tiles, conv_thin16_kernel)
        fg = g.standard_normal((N, c, H, W)).astype(np.float32)
        fe = g.standard_normal((N, c, H, W)).astype(np.float32)
        w = (g.standard_normal((1, 2 * c, 3, 3)) / math.sqrt(18 * c)).astype(np.float32)
        bias = T(g.standard_normal(1).astype(np.float32) * 0.1, dev)
        sc = T((1 + 0.3 * g.standard_normal((N, 2 * c))).astype(np.float32), dev)
        sh = T((0.3 * g.standard_normal((N, 2 * c))).astype(np.float32), dev)
        fgt, fet = K.nchw_to_nhwc(T(fg, dev), dtype), K.nchw_to_nhwc(T(fe, dev), dtype)
        wp = K.pack_conv_weight(T(w, dev), out_dtype=dtype)
        # two launches: normalised copy, then the conv
        nrm = torch.zeros((N, H, W, 2 * c), dtype=dtype, device=dev)
        K.affine_apply(nrm, 2 * c, fgt, c, sc, sh, N, H * W, c, code, other=fet, ld_other=c)
        ref = torch.zeros((N, 1, H, W), device=dev)
        common = dict(n=N, h=H, w=W, out_h=H, out_w=W, weight=wp, cout=1, kh=3, kw=3, pad=1, bias=bias,
                      act=K.ACT_RELU_TANH, ld_out=0, out_layout=K.OUT_NCHW, out_dtype=K.VT_F32, dtype=code)
        K.conv2d(src0=nrm, c0=2 * c, ld0=2 * c, out=ref, **common)
        # one launch
        out = torch.zeros_like(ref)
        d = K.make_conv_desc(src0=fgt, c0=c, ld0=c, src1=fet, c1=c, ld1=c, in_scale=sc, in_shift=sh, in_absdiff=1, out=out,
                             **common)
        assert lib.vt_conv2d_tile(C.byref(d)) // 100000000 == 6
        assert lib.vt_conv2d(C.byref(d), K._stream(out)) == 0, lib.vt_last_error()
        assert float(ref.abs().max()) > 0.05 and torch.equal(out, ref), (N, c, H, W)
        # the affine alone (one source) through the same loader
        ref1, out1 = torch.zeros_like(ref), torch.zeros_like(ref)
        nrm1 = torch.zeros((N, H, W, c), dtype=dtype, device=dev)
        w1 = K.pack_conv_weight(T(w[:, :c].copy(), dev), out_dtype=dtype)
        K.affine_apply(nrm1, c, fgt, c, sc[:, :c].contiguous(), sh[:, :c].contiguous(), N, H * W, c, code)
        K.conv2d(src0=nrm1, c0=c, ld0=c, out=ref1, **{**common, "weight": w1})
        K.conv2d(src0=fgt, c0=c, ld0=c, in_scale=sc[:, :c].contiguous(), in_shift=sh[:, :c].contiguous(), out=out1,
                 **{**common, "weight": w1})
        assert torch.equal(out1, ref1)
    # three planar outputs (27 virtual channels) have no prologue instance: refused, not silently wrong
    w3 = K.pack_conv_weight(T((g.standard_normal((3, 2 * c, 3, 3)) / 10).astype(np.float32), dev), out_dtype=dtype)
    o3 = torch.zeros((N, 3, H, W), device=dev)
    d = K.make_conv_desc(src0=fgt, c0=c, ld0=c, src1=fet, c1=c, ld1=c, in_absdiff=1, out=o3,
                         **{**common, "weight": w3, "cout": 3, "bias": None})
    assert lib.vt_conv2d(C.byref(d), K._stream(o3)) == 2   # VT_ERR_UNSUPPORTED


@pytest.mark.parametrize("hint", [128128, 128064, 128032, 128016, 64064, 64128, 32064])
def test_conv_every_tile(dev, hint):
    for dtype in (torch.float32, torch.bfloat16):
        t = F32_TOL if dtype == torch.float32 else 8e-3
        assert _conv_case(dev, dtype, 1, 48, 10, 13, 136, 3, 1, 1, 1, act=K.ACT_LRELU, hint=hint, resid=True) < t


P = 100000000  # tile-code digit of the patch-resident kernel


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_direct_to_lds_and_patch_kernels(dev, dtype):
    """cin % 64 == 0 (bf16) / % 32 (fp32) makes a conv eligible for the buffer_load->LDS loaders;
    3x3 stride-1 pad==dil additionally for the patch-resident kernel.  Odd sizes exercise the
    zero-fill padding (out-of-range buffer offsets), 2-D tile edges, batches and split-K."""
    t = F32_TOL if dtype == torch.float32 else 8e-3
    L = K.ACT_LRELU
    # patch kernel, auto plan (small image => 128x64 tiles + split-K over channel chunks)
    assert _conv_case(dev, dtype, 2, 128, 19, 37, 64, 3, 1, 1, 1, act=L, resid=True, ws=True, expect_kind=1) < t
    # dilated convs run the 1-D kernel by default (measured faster); the patch instances are forced
    # dilated convs: deep-ring 128x128 patch instance, one channel chunk per K slice (the trunk)
    assert _conv_case(dev, dtype, 1, 128, 9, 21, 136, 3, 1, 2, 2, act=L, ws=True, expect_kind=1) < t
    assert _conv_case(dev, dtype, 2, 128, 11, 17, 128, 3, 1, 4, 4, resid=True, ws=True, expect_kind=1) < t
    assert _conv_case(dev, dtype, 1, 128, 11, 17, 64, 3, 1, 4, 4, ws=True, expect_kind=0) < t   # cout 64: 1-D
    assert _conv_case(dev, dtype, 1, 128, 11, 17, 128, 3, 1, 4, 4, ws=False, expect_kind=0) < t  # no workspace: 1-D
    assert _conv_case(dev, dtype, 1, 192, 8, 8, 1, 3, 1, 1, 1, act=K.ACT_RELU_TANH, planar=True, ws=True,
                      expect_kind=1) < t                                                               # mask conv
    # every compiled patch tile, forced by hint (S=0: auto split; S=2 forced)
    for hint in (P + 256128, P + 256064, P + 128064, P + 128016, P + 2000000 + 128064, P + 2000000 + 256128,
                 P + 128128, P + 1000000 + 128128, P + 2000000 + 128128):
        assert _conv_case(dev, dtype, 1, 128, 21, 35, 136, 3, 1, 1, 1, act=L, hint=hint, resid=True, ws=True,
                          seed=hint % 97) < t, hint
    # 1-D direct-to-LDS kernels (patch disabled with P=2), incl. stride 2 and a 1x1 conv, with split-K
    for hint in (2 * P + 128128, 2 * P + 64064, 2 * P + 4000000 + 64064, 2 * P + 128064, 2 * P + 64128):
        assert _conv_case(dev, dtype, 2, 64, 13, 10, 136, 3, 1, 1, 1, act=L, hint=hint, resid=True, ws=True,
                          expect_kind=0) < t, hint
    assert _conv_case(dev, dtype, 1, 128, 13, 18, 64, 3, 2, 1, 1, act=L, ws=True, expect_kind=0) < t   # stride 2
    assert _conv_case(dev, dtype, 2, 64, 9, 7, 3, 1, 1, 0, 1, planar=True, resid=True, ws=True) < t    # 1x1


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_fused_torgb(dev, dtype):
    """StyledConv + ToRGB in one launch (vt_conv_desc.rgb_*): rgb = W_rgb . act(conv) + bias + skip, for
    every kernel family / wave layout that can hold all channels in one tile; refused otherwise."""
    g = np.random.default_rng(12)
    tol = F32_TOL if dtype == torch.float32 else 1.5e-2
    for cin, cout, H, W, hint in [(64, 128, 19, 37, P + 1000000 + 256128),      # patch, 8 waves, channels over 2 waves
                                  (64, 64, 21, 35, P + 1000000 + 256064),
                                  (64, 64, 9, 40, P + 1000000 + 128064),
                                  (32, 32, 17, 33, 0),                          # persistent 32->32 kernel (bf16) / tile kernel
                                  (64, 64, 23, 41, 0),                          # persistent 64->64 kernel (bf16): halves meet in LDS
                                  (64, 128, 12, 20, 2 * P + 1000000 + 128128)]:  # 1-D direct-to-LDS, 2x2 waves
        x = g.standard_normal((2, cin, H, W)).astype(np.float32)
        w = (g.standard_normal((cout, cin, 3, 3)) / math.sqrt(9 * cin)).astype(np.float32)
        b = g.standard_normal(cout).astype(np.float32)
        wr = (g.standard_normal((3, cout, 1, 1)) / math.sqrt(cout)).astype(np.float32)
        br = g.standard_normal(3).astype(np.float32)
        skip = g.standard_normal((2, 3, H, W)).astype(np.float32)
        xt = K.nchw_to_nhwc(T(x, dev), dtype)
        wp = K.pack_conv_weight(T(w, dev), out_dtype=dtype)
        wrp = K.pack_conv_weight(T(wr, dev), out_dtype=dtype)
        xq = xt.float().cpu().permute(0, 3, 1, 2).numpy()
        wq = wp.float().cpu().numpy().reshape(cout, 3, 3, cin).transpose(0, 3, 1, 2)
        wrq = wrp.float().cpu().numpy().reshape(3, 1, 1, cout).transpose(0, 3, 1, 2)
        y_ref = O.leaky_relu(O.conv2d(xq, wq, b, 1, 1, 1), 0.2) * np.float32(2 ** 0.5)
        rgb_ref = O.conv2d(y_ref, wrq, br) + skip
        out = torch.zeros((2, H, W, cout), dtype=dtype, device=dev)
        rgb = T(skip.copy(), dev)
        K.conv2d(src0=xt, c0=cin, ld0=cin, n=2, h=H, w=W, out_h=H, out_w=W, weight=wp, cout=cout, kh=3, kw=3, pad=1,
                 bias=T(b, dev), act=K.ACT_LRELU, gain=2 ** 0.5, out=out, ld_out=cout, dtype=K.dt_code(dtype),
                 tile_hint=hint, rgb_weight=wrp, rgb_bias=T(br, dev), rgb_resid=rgb, rgb_out=rgb)
        assert rel_err(out.float().cpu().permute(0, 3, 1, 2).numpy(), y_ref) < (F32_TOL if dtype == torch.float32 else 8e-3)
        assert rel_err(rgb.cpu().numpy(), rgb_ref) < tol, (cin, cout, hint)
        if (cin, cout) == (32, 32) and dtype == torch.bfloat16:
            # vt_conv_desc.rgb_only (ABI 4): the last level's activation is not stored -- same image, `out` untouched
            out2 = torch.full((2, H, W, cout), 7.0, dtype=dtype, device=dev)
            rgb2 = T(skip.copy(), dev)
            K.conv2d(src0=xt, c0=cin, ld0=cin, n=2, h=H, w=W, out_h=H, out_w=W, weight=wp, cout=cout, kh=3, kw=3, pad=1,
                     bias=T(b, dev), act=K.ACT_LRELU, gain=2 ** 0.5, out=out2, ld_out=cout, dtype=K.dt_code(dtype),
                     tile_hint=hint, rgb_weight=wrp, rgb_bias=T(br, dev), rgb_resid=rgb2, rgb_out=rgb2, rgb_only=1)
            assert torch.equal(rgb2, rgb) and bool((out2 == 7.0).all())
        elif (cin, cout) == (64, 64):
            with pytest.raises(Exception, match="rgb_only"):
                K.conv2d(src0=xt, c0=cin, ld0=cin, n=2, h=H, w=W, out_h=H, out_w=W, weight=wp, cout=cout, kh=3, kw=3,
                         pad=1, out=out, ld_out=cout, dtype=K.dt_code(dtype), tile_hint=hint, rgb_weight=wrp,
                         rgb_bias=T(br, dev), rgb_resid=rgb, rgb_out=rgb, rgb_only=1)
    # more channels than one tile: refused (the engine then issues ToRGB as its own conv)
    xt = K.nchw_to_nhwc(T(g.standard_normal((1, 64, 8, 8)).astype(np.float32), dev), dtype)
    wp = K.pack_conv_weight(T((g.standard_normal((256, 64, 3, 3)) / 24).astype(np.float32), dev), out_dtype=dtype)
    wrp = K.pack_conv_weight(T((g.standard_normal((3, 256, 1, 1)) / 16).astype(np.float32), dev), out_dtype=dtype)
    rgb = torch.zeros((1, 3, 8, 8), device=dev)
    with pytest.raises(Exception, match="fused ToRGB"):
        K.conv2d(src0=xt, c0=64, ld0=64, n=1, h=8, w=8, out_h=8, out_w=8, weight=wp, cout=256, kh=3, kw=3, pad=1,
                 out=torch.zeros((1, 8, 8, 256), dtype=dtype, device=dev), ld_out=256, dtype=K.dt_code(dtype),
                 rgb_weight=wrp, rgb_out=rgb)


def test_conv_c32_persistent_kernel(dev, monkeypatch):
    """3x3 32->32 bf16 (the 1024^2 level) runs the persistent register-weight kernel: several tiles per
    workgroup (double-buffered patches), image borders, batch, residual epilogue."""
    t = 8e-3
    import ctypes
    from vtoonify_amd import _lib
    monkeypatch.setenv("VT_C32_BLOCKS", "3")     # 2 x 3 x 4 = 24 tiles over 3 workgroups
    assert _conv_case(dev, torch.bfloat16, 2, 32, 37, 50, 32, 3, 1, 1, 1, act=K.ACT_LRELU) < t
    assert _conv_case(dev, torch.bfloat16, 2, 32, 21, 18, 32, 3, 1, 1, 1, act=K.ACT_LRELU, resid=True) < t   # generic
    monkeypatch.delenv("VT_C32_BLOCKS")
    assert _conv_case(dev, torch.bfloat16, 1, 32, 16, 16, 32, 3, 1, 1, 1) < t
    x = K.nchw_to_nhwc(torch.zeros(1, 32, 16, 16, device=dev), torch.bfloat16)
    d = K.make_conv_desc(src0=x, c0=32, ld0=32, n=1, h=16, w=16, out_h=16, out_w=16, weight=x, cout=32, kh=3, kw=3,
                         pad=1, out=x, ld_out=32, dtype=K.VT_BF16)
    assert _lib.lib().vt_conv2d_tile(ctypes.byref(d)) // 100000000 == 3
    # cout = 32k (the encoder's 32 -> 128 conv): one group of 32 output channels per blockIdx.y, same kernel
    for cout in (64, 128):
        monkeypatch.setenv("VT_C32_BLOCKS", "2")
        assert _conv_case(dev, torch.bfloat16, 2, 32, 21, 34, cout, 3, 1, 1, 1, act=K.ACT_LRELU, expect_kind=3) < t
        monkeypatch.delenv("VT_C32_BLOCKS")
    # fp32 (parity mode) and dilated / strided 32->32 convs stay on the generic kernels
    assert _conv_case(dev, torch.float32, 1, 32, 19, 21, 32, 3, 1, 1, 1, act=K.ACT_LRELU) < F32_TOL
    assert _conv_case(dev, torch.bfloat16, 1, 32, 19, 21, 32, 3, 2, 1, 1) < t


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_emits_instnorm_records(dev, dtype):
    """vt_conv_desc.stats_part: the split-K reduce pass writes the InstanceNorm chunk records of the
    tensor it stores (else the statistics launch is appended) -- AdaIN from those records is
    BIT-identical to AdaIN from the stand-alone statistics pass over the stored tensor."""
    import ctypes as C
    from vtoonify_amd import _lib
    g = np.random.default_rng(21)
    N, Ci, H, W, Co = 2, 128, 9, 11, 48
    x = g.standard_normal((N, Ci, H, W)).astype(np.float32)
    r = g.standard_normal((N, Co, H, W)).astype(np.float32)
    w = (g.standard_normal((Co, Ci, 3, 3)) / 30).astype(np.float32)
    b = g.standard_normal(Co).astype(np.float32)
    gb = g.standard_normal((N, 2 * Co)).astype(np.float32)
    code = K.dt_code(dtype)
    xt, rt = K.nchw_to_nhwc(T(x, dev), dtype), K.nchw_to_nhwc(T(r, dev), dtype)
    wp = K.pack_conv_weight(T(w, dev), out_dtype=dtype)
    wsk = torch.zeros(8 << 20, dtype=torch.float32, device=dev)
    lib = _lib.lib()
    hw = H * W
    gbt = T(gb, dev)
    for ws in (wsk, None):           # split-K (records from the reduce pass) / one pass (launch appended)
        out = torch.zeros((N, H, W, Co), dtype=dtype, device=dev)
        part = torch.zeros(K.instnorm_ws_bytes(N, hw, Co), dtype=torch.uint8, device=dev)
        d = K.make_conv_desc(src0=xt, c0=Ci, ld0=Ci, n=N, h=H, w=W, out_h=H, out_w=W, weight=wp, cout=Co, kh=3, kw=3,
                             pad=1, bias=T(b, dev), act=K.ACT_LRELU, alpha=0.7, beta=0.7, resid=rt, ld_res=Co,
                             out=out, ld_out=Co, dtype=code, splitk_ws=ws, stats_part=part)
        if ws is not None:
            assert (lib.vt_conv2d_tile(C.byref(d)) // 1000000) % 100 > 1     # really split
        assert lib.vt_conv2d(C.byref(d), K._stream(xt)) == 0, lib.vt_last_error()
        a = torch.zeros_like(out)
        assert lib.vt_instnorm_apply_stats(a.data_ptr(), Co, out.data_ptr(), Co, N, hw, Co, gbt.data_ptr(), 2 * Co,
                                           part.data_ptr(), code, K._stream(xt)) == 0
        ref = torch.zeros_like(out)
        part2 = torch.zeros_like(part)
        assert lib.vt_instnorm_apply(ref.data_ptr(), Co, out.data_ptr(), Co, N, hw, Co, gbt.data_ptr(), 2 * Co,
                                     part2.data_ptr(), code, K._stream(xt)) == 0
        assert torch.equal(part, part2)
        assert torch.equal(a, ref)
        y = O.conv2d(xt.float().cpu().permute(0, 3, 1, 2).numpy(), wp_ref(w, dtype), b, 1, 1, 1)
        y = np.where(y > 0, y, 0.2 * y) * 0.7 + 0.7 * rt.float().cpu().permute(0, 3, 1, 2).numpy()
        assert rel_err(out.float().cpu().permute(0, 3, 1, 2).numpy(), y) < (F32_TOL if dtype == torch.float32 else 2e-2)


def wp_ref(w, dtype):
    return w if dtype == torch.float32 else torch.from_numpy(w).to(torch.bfloat16).float().numpy()


def test_conv_batch_aware_tiles(dev, monkeypatch):
    """Plan choices that look at the batch (vt_conv2d_tile, no compute on CPU): a batch that fills the GPU with
    256-pixel x 128-channel patch tiles gets them -- where one frame alone takes 128 x 64 tiles of the same kernel the bits
    are the same (checked on the GPU), where it takes the whole-K kernels the results agree to rounding and
    VT_BATCH_EXACT=1 restores the per-image choice."""
    import ctypes
    from vtoonify_amd import _lib
    lib = _lib.lib()

    def code(N, cin, H, W, cout, stream=False, exact=None):
        if exact is None:
            monkeypatch.delenv("VT_BATCH_EXACT", raising=False)
        else:
            monkeypatch.setenv("VT_BATCH_EXACT", exact)
        x = torch.zeros((1,), dtype=torch.bfloat16, device=dev)      # never dereferenced by the query
        d = K.make_conv_desc(src0=x, c0=cin, ld0=cin, n=N, h=H, w=W, out_h=H, out_w=W, weight=x, cout=cout, kh=3, kw=3,
                             pad=1, out=x, ld_out=cout, dtype=K.VT_BF16)
        if stream:
            d.weight_stream = x.data_ptr()
        d.splitk_ws, d.splitk_ws_bytes = x.data_ptr(), 1 << 40
        return lib.vt_conv2d_tile(ctypes.byref(d))

    assert code(1, 256, 128, 128, 256) == 101128064                    # one frame: 128 x 64 patch tiles, no K split
    assert code(4, 256, 128, 128, 256) == 101256128                    # four: 256 x 128 tiles of the same kernel
    assert code(4, 256, 128, 128, 256, exact="1") == 101256128         # (same bits: allowed under VT_BATCH_EXACT)
    assert code(1, 512, 64, 64, 512, stream=True) // 100000000 in (4, 8)   # whole-K kernels for one frame ...
    assert code(4, 512, 64, 64, 512, stream=True) == 101256128             # ... patch tiles for a batch that fills the GPU
    assert code(4, 512, 64, 64, 512, stream=True, exact="1") // 100000000 in (4, 8)
    assert code(4, 512, 32, 32, 512, stream=True) == 101256032         # the 32 x 32 trunk: 32-channel tiles, one per CU
    assert code(2, 512, 32, 32, 512, stream=True) // 100000000 == 8    # ... two frames do not fill the GPU: weight-stationary
    assert code(4, 512, 32, 32, 512, stream=True, exact="1") // 100000000 == 8
    monkeypatch.delenv("VT_BATCH_EXACT", raising=False)
    if dev.type != "cuda":
        return
    # GPU: the 256 x 128 tiles of a batch against the same frames one at a time
    g = np.random.default_rng(9)
    for cin, H, W, cout, same_bits in ((256, 128, 128, 256, True), (512, 64, 64, 512, False)):
        x = g.standard_normal((4, cin, H, W)).astype(np.float32)
        w = (g.standard_normal((cout, cin, 3, 3)) / math.sqrt(cin * 9)).astype(np.float32)
        xt = K.nchw_to_nhwc(T(x, dev), torch.bfloat16)
        wp = K.pack_conv_weight(T(w, dev), out_dtype=torch.bfloat16)
        wst = K.conv_weight_stream(wp)
        ws = torch.zeros(64 << 20, dtype=torch.float32, device=dev)

        def run(xb):
            out = torch.zeros((xb.shape[0], H, W, cout), dtype=torch.bfloat16, device=dev)
            d = K.make_conv_desc(src0=xb, c0=cin, ld0=cin, n=xb.shape[0], h=H, w=W, out_h=H, out_w=W, weight=wp, cout=cout,
                                 kh=3, kw=3, pad=1, act=K.ACT_LRELU, out=out, ld_out=cout, dtype=K.VT_BF16, splitk_ws=ws)
            if wst is not None:
                d.weight_stream = wst.data_ptr()
            assert lib.vt_conv2d(ctypes.byref(d), K._stream(xb)) == 0, lib.vt_last_error()
            return out
        full = run(xt)
        alone = torch.cat([run(xt[i:i + 1].contiguous()) for i in range(4)])
        if same_bits:
            assert torch.equal(full, alone)
        else:
            assert rel_err(full.float().cpu().numpy(), alone.float().cpu().numpy()) < 4e-3
            monkeypatch.setenv("VT_BATCH_EXACT", "1")
            assert torch.equal(run(xt), alone)
            monkeypatch.delenv("VT_BATCH_EXACT")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_patch_pipelined_dilated(dev, dtype):
    """conv_patchp_kernel<.., DIL = 2> (round 5): the dilation-2 convs of the AdaResBlocks on the 256-pixel x 32-channel pipelined
    tiles (20 x 20-pixel patches).  Against the oracle: several chunks of K, ragged tiles, a batch, residual + LeakyReLU; and the
    plan query picks it for a batch of the 32 x 32 trunk (not for dilation 4, not under VT_BATCH_EXACT)."""
    t = F32_TOL if dtype == torch.float32 else 8e-3
    L = K.ACT_LRELU
    for N, cin, H, W, cout, resid in ((2, 128, 20, 24, 32, False), (1, 64, 33, 17, 64, True), (3, 192, 16, 16, 40, True)):
        assert _conv_case(dev, dtype, N, cin, H, W, cout, 3, 1, 2, 2, act=L, resid=resid, hint=P + 256032, expect_kind=1) < t, \
            (N, cin, H, W, cout)
    if dtype == torch.bfloat16:
        import ctypes
        from vtoonify_amd import _lib
        x = torch.zeros((1,), dtype=torch.bfloat16, device=dev)
        for dil, want in ((1, 101256032), (2, 101256032), (4, None)):
            d = K.make_conv_desc(src0=x, c0=512, ld0=512, n=4, h=32, w=32, out_h=32, out_w=32, weight=x, cout=512, kh=3, kw=3,
                                 pad=dil, dil=dil, out=x, ld_out=512, dtype=K.VT_BF16)
            d.weight_stream = x.data_ptr()
            d.splitk_ws, d.splitk_ws_bytes = x.data_ptr(), 1 << 40
            code = _lib.lib().vt_conv2d_tile(ctypes.byref(d))
            assert (code == want) if want else (code // 100000000 == 8), (dil, code)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_patch_pipelined_equals_per_tap(dev, dtype, monkeypatch):
    """conv_patchp_kernel (csrc/conv_patch_pipe.hpp: fragments of tap s+1 read before the barrier of tap s, 4-deep weight
    ring, one patch piece per tap, LDS-DMA between the two halves of a tap) sums K in the order of conv_patch_kernel
    ([chunk][tap][half]): the 256-pixel tiles must give the SAME BITS on both -- several chunks (the double-buffered
    patch), two concatenated sources, ragged tile edges, a batch, split-K slices, residual + activation epilogue."""
    unit = 64 if dtype == torch.bfloat16 else 32
    g = np.random.default_rng(31)
    # (the 32-channel tiles -- deep weight ring, three patch pieces per tap -- have no per-tap twin: the K order does not
    # depend on the tile width, so their reference is the 128-channel per-tap instance)
    for N, c0, c1, H, W, cout, hint, ref_hint in ((2, 3 * unit, 0, 21, 35, 136, P + 256128, P + 256128),
                                                  (1, unit, 2 * unit, 17, 16, 128, P + 256128, P + 256128),
                                                  (1, 2 * unit, 0, 16, 33, 72, P + 256064, P + 256064),
                                                  (1, 4 * unit, 0, 19, 18, 136, P + 2000000 + 256128, P + 2000000 + 256128),
                                                  (2, 3 * unit, 0, 21, 35, 136, P + 256032, P + 256128),
                                                  (1, unit, 2 * unit, 33, 17, 40, P + 256032, P + 256128)):
        cin = c0 + c1
        x = g.standard_normal((N, cin, H, W)).astype(np.float32)
        w = (g.standard_normal((cout, cin, 3, 3)) / math.sqrt(cin * 9)).astype(np.float32)
        b = g.standard_normal(cout).astype(np.float32)
        xa = K.nchw_to_nhwc(T(x[:, :c0], dev), dtype)
        xb = K.nchw_to_nhwc(T(x[:, c0:], dev), dtype) if c1 else None
        wp = K.pack_conv_weight(T(w, dev), out_dtype=dtype)
        ldo = (cout + 7) // 8 * 8
        r = K.nchw_to_nhwc(T(g.standard_normal((N, cout, H, W)).astype(np.float32), dev), dtype, ld_out=ldo)
        ws = torch.zeros(8 << 20, dtype=torch.float32, device=dev)

        def run(hint):
            out = torch.zeros((N, H, W, ldo), dtype=dtype, device=dev)
            kw = dict(src1=xb, c1=c1, ld1=c1) if c1 else {}
            K.conv2d(src0=xa, c0=c0, ld0=c0, n=N, h=H, w=W, out_h=H, out_w=W, weight=wp, cout=cout, kh=3, kw=3, pad=1,
                     bias=T(b, dev), act=K.ACT_LRELU, gain=2 ** 0.5, alpha=0.5, beta=0.25, resid=r, ld_res=ldo, out=out,
                     ld_out=ldo, dtype=K.dt_code(dtype), tile_hint=hint, splitk_ws=ws, **kw)
            return out
        monkeypatch.setenv("VT_PATCH_PIPE", "0")
        ref = run(ref_hint)
        monkeypatch.delenv("VT_PATCH_PIPE")
        got = run(hint)
        assert torch.equal(got, ref), (N, c0, c1, H, W, cout, hint)
        xq = torch.cat([xa] + ([xb] if c1 else []), dim=3).float().cpu().permute(0, 3, 1, 2).numpy()
        wq = wp.float().cpu().numpy().reshape(cout, 3, 3, cin).transpose(0, 3, 1, 2)
        want = O.leaky_relu(O.conv2d(xq, wq, b, 1, 1, 1), 0.2) * np.float32(2 ** 0.5) * 0.5 + \
            0.25 * r.float().cpu().permute(0, 3, 1, 2)[:, :cout].numpy()
        assert rel_err(got.float().cpu().permute(0, 3, 1, 2)[:, :cout].numpy(), want) < (F32_TOL if dtype == torch.float32 else 8e-3)


def test_conv_patch_chunk_equals_pipelined(dev, monkeypatch):
    """conv_patchc_kernel (csrc/conv_patch_chunk.hpp, round 6): the 32-channel tiles with all nine taps of a chunk resident and ONE
    barrier per chunk.  Same tile, same loader addresses, same K order as the tap-granular pipeline (VT_PATCH_PIPE=1 ->
    conv_patchp_kernel<T,16,32,8,1,8>): the SAME BITS -- one chunk, many chunks (both chunk buffers in use), two concatenated
    sources whose seam is a chunk boundary, ragged tile edges, a batch, more channel tiles than one, split-K slices (uneven), the
    lean epilogue (no residual) and the general one."""
    dtype = torch.bfloat16
    g = np.random.default_rng(61)
    # dil = 2 (conv_patchc_kernel<.., DIL = 2>: tiles over the four sub-images of an image, dense 18 x 18 patches of a sub-image)
    # against conv_patchp_kernel<.., DIL = 2> (20 x 20 patches of the full image): even and odd sizes, sub-images smaller and
    # larger than a tile
    for N, c0, c1, H, W, cout, split, resid, dil in ((1, 64, 0, 16, 16, 32, 0, False, 1), (2, 320, 0, 21, 35, 72, 0, True, 1),
                                                      (1, 64, 128, 33, 17, 40, 0, True, 1), (3, 256, 0, 32, 32, 64, 0, False, 1),
                                                      (1, 448, 0, 19, 18, 96, 3000000, True, 1),
                                                      (4, 512, 0, 32, 32, 64, 0, True, 1),
                                                      (4, 256, 0, 32, 32, 64, 0, True, 2), (2, 128, 0, 21, 35, 40, 0, False, 2),
                                                      (1, 64, 64, 45, 50, 32, 0, True, 2), (1, 192, 0, 18, 32, 72, 2000000, True, 2)):
        cin = c0 + c1
        x = g.standard_normal((N, cin, H, W)).astype(np.float32)
        w = (g.standard_normal((cout, cin, 3, 3)) / math.sqrt(cin * 9)).astype(np.float32)
        b = g.standard_normal(cout).astype(np.float32)
        xa = K.nchw_to_nhwc(T(x[:, :c0], dev), dtype)
        xb = K.nchw_to_nhwc(T(x[:, c0:], dev), dtype) if c1 else None
        wp = K.pack_conv_weight(T(w, dev), out_dtype=dtype)
        ldo = (cout + 7) // 8 * 8
        r = K.nchw_to_nhwc(T(g.standard_normal((N, cout, H, W)).astype(np.float32), dev), dtype, ld_out=ldo) if resid else None
        ws = torch.zeros(8 << 20, dtype=torch.float32, device=dev)

        def run():
            out = torch.zeros((N, H, W, ldo), dtype=dtype, device=dev)
            kw = dict(src1=xb, c1=c1, ld1=c1) if c1 else {}
            if resid:
                kw.update(alpha=0.5, beta=0.25, resid=r, ld_res=ldo)
            K.conv2d(src0=xa, c0=c0, ld0=c0, n=N, h=H, w=W, out_h=H, out_w=W, weight=wp, cout=cout, kh=3, kw=3, pad=dil, dil=dil,
                     bias=T(b, dev), act=K.ACT_LRELU, gain=2 ** 0.5, out=out, ld_out=ldo, dtype=K.dt_code(dtype),
                     tile_hint=P + split + 256032, splitk_ws=ws, **kw)
            return out
        monkeypatch.setenv("VT_PATCH_PIPE", "1")
        ref = run()
        monkeypatch.delenv("VT_PATCH_PIPE")
        got = run()
        assert torch.equal(got, ref), (N, c0, c1, H, W, cout, split, dil)
        xq = torch.cat([xa] + ([xb] if c1 else []), dim=3).float().cpu().permute(0, 3, 1, 2).numpy()
        wq = wp.float().cpu().numpy().reshape(cout, 3, 3, cin).transpose(0, 3, 1, 2)
        want = O.leaky_relu(O.conv2d(xq, wq, b, 1, dil, dil), 0.2) * np.float32(2 ** 0.5)
        if resid:
            want = want * 0.5 + 0.25 * r.float().cpu().permute(0, 3, 1, 2)[:, :cout].numpy()
        assert rel_err(got.float().cpu().permute(0, 3, 1, 2)[:, :cout].numpy(), want) < 8e-3


def test_conv_stride2_by_input_parity(dev):
    """conv_patchs2_kernel (csrc/conv_patch_s2.hpp, round 6): the encoder's stride-2 3x3 convs (model/vtoonify.py:167-176) as four
    dense convs on the parity sub-images of the input, patch-resident.  Against the oracle: one and several chunks, ragged tiles
    in both directions (output sizes that are not multiples of 16), a ragged channel tile, a batch, LeakyReLU with and without a
    residual (general / lean epilogue), planar fp32 output; the frame of a batch equals the frame alone (no K split, the kernel
    choice is per image from 128 tiles up); and the plan query picks it for the 256^2 and 128^2 stages of the encoder, for the
    64^2 one only in a batch."""
    import ctypes
    from vtoonify_amd import _lib
    S2 = 700000000 + 256064
    dt, L = torch.bfloat16, K.ACT_LRELU
    for N, cin, H, W, cout, resid, planar in ((1, 64, 32, 32, 64, False, False), (2, 128, 40, 72, 64, True, False),
                                              (1, 192, 36, 20, 72, False, False), (3, 64, 16, 16, 40, True, False),
                                              (1, 128, 66, 34, 128, False, True)):
        for hint in (S2, S2 - 32):   # 64- and 32-channel tiles
            assert _conv_case(dev, dt, N, cin, H, W, cout, 3, 2, 1, 1, act=L, resid=resid, planar=planar, hint=hint) < 8e-3, \
                (N, cin, H, W, cout, hint)
    x = torch.zeros((1,), dtype=dt, device=dev)
    for cin, cout, hw, want1, want4 in ((128, 256, 256, 7, 7), (256, 512, 128, 7, 7), (512, 512, 64, 2, 7)):
        for n, want in ((1, want1), (4, want4)):   # (the deepest stage: 32 tiles per image -- the 1-D tiles alone, this kernel in a batch)
            d = K.make_conv_desc(src0=x, c0=cin, ld0=cin, n=n, h=hw, w=hw, out_h=hw // 2, out_w=hw // 2, weight=x, cout=cout, kh=3,
                                 kw=3, stride=2, pad=1, out=x, ld_out=cout, dtype=K.VT_BF16)
            d.splitk_ws, d.splitk_ws_bytes = x.data_ptr(), 1 << 40
            assert _lib.lib().vt_conv2d_tile(ctypes.byref(d)) // 100000000 == want, (cin, cout, hw, n)
    # a frame inside a batch == the frame alone
    g = np.random.default_rng(7)
    xs = g.standard_normal((3, 128, 32, 48)).astype(np.float32)
    w = (g.standard_normal((64, 128, 3, 3)) / 34.0).astype(np.float32)
    wp = K.pack_conv_weight(T(w, dev), out_dtype=dt)

    def run(xn):
        xt = K.nchw_to_nhwc(T(xn, dev), dt)
        out = torch.zeros((xn.shape[0], 16, 24, 64), dtype=dt, device=dev)
        K.conv2d(src0=xt, c0=128, ld0=128, n=xn.shape[0], h=32, w=48, out_h=16, out_w=24, weight=wp, cout=64, kh=3, kw=3, stride=2,
                 pad=1, out=out, ld_out=64, dtype=K.VT_BF16, tile_hint=S2)
        return out
    full = run(xs)
    for i in range(3):
        assert torch.equal(run(xs[i:i + 1])[0], full[i])
    # ... and the 32-channel tiles give the bits of the 64-channel ones (same K order): the width may follow the batch
    xt = K.nchw_to_nhwc(T(xs, dev), dt)
    o32 = torch.zeros((3, 16, 24, 64), dtype=dt, device=dev)
    K.conv2d(src0=xt, c0=128, ld0=128, n=3, h=32, w=48, out_h=16, out_w=24, weight=wp, cout=64, kh=3, kw=3, stride=2, pad=1, out=o32,
             ld_out=64, dtype=K.VT_BF16, tile_hint=S2 - 32)
    assert torch.equal(o32, full)


def test_conv_patch_persistent_equals_one_workgroup_per_tile(dev, monkeypatch):
    """conv_patchq_kernel (csrc/conv_patch_persist.hpp): persistent workgroups walk contiguous ranges of tiles and the loader
    prefetches across tile boundaries (the successor's first patch and taps land during the epilogue).  Same K order, same lean
    epilogue -> the SAME BITS as one workgroup per tile (VT_PATCH_PIPE=1).  VT_PATCHW_WGS = few workgroups so that small
    convolutions have more tiles than workgroups: ranges that cross into the next channel tile (tables reloaded), uneven ranges,
    several chunks, two sources, ragged edges, a batch, residual + LeakyReLU, the fused ToRGB (its LDS exchange lives behind the
    patch here), 128- and 64-channel tiles."""
    dtype = torch.bfloat16
    g = np.random.default_rng(43)
    for N, c0, c1, H, W, cout, hint, wgs, rgb in ((2, 192, 0, 21, 35, 136, P + 256128, 3, False),
                                                  (1, 64, 128, 33, 40, 128, P + 256128, 4, True),
                                                  (3, 128, 0, 17, 31, 72, P + 256064, 5, False),
                                                  (1, 128, 0, 40, 50, 64, P + 256064, 2, True)):
        cin = c0 + c1
        x = g.standard_normal((N, cin, H, W)).astype(np.float32)
        w = (g.standard_normal((cout, cin, 3, 3)) / math.sqrt(cin * 9)).astype(np.float32)
        b = g.standard_normal(cout).astype(np.float32)
        xa = K.nchw_to_nhwc(T(x[:, :c0], dev), dtype)
        xb = K.nchw_to_nhwc(T(x[:, c0:], dev), dtype) if c1 else None
        wp = K.pack_conv_weight(T(w, dev), out_dtype=dtype)
        ldo = cout
        r = K.nchw_to_nhwc(T(g.standard_normal((N, cout, H, W)).astype(np.float32), dev), dtype, ld_out=ldo)
        rgbw = K.pack_conv_weight(T((g.standard_normal((3, cout, 1, 1)) / 8).astype(np.float32), dev), out_dtype=dtype)
        skip = T(g.standard_normal((N, 3, H, W)).astype(np.float32), dev)
        rgbb = T(g.standard_normal(3).astype(np.float32), dev)

        def run():
            out = torch.zeros((N, H, W, ldo), dtype=dtype, device=dev)
            rgb_out = torch.zeros((N, 3, H, W), dtype=torch.float32, device=dev)
            kw = dict(src1=xb, c1=c1, ld1=c1) if c1 else {}
            if rgb:
                kw.update(rgb_weight=rgbw, rgb_bias=rgbb, rgb_resid=skip, rgb_out=rgb_out)
            else:
                kw.update(resid=r, ld_res=ldo, beta=0.25)
            K.conv2d(src0=xa, c0=c0, ld0=c0, n=N, h=H, w=W, out_h=H, out_w=W, weight=wp, cout=cout, kh=3, kw=3, pad=1,
                     bias=T(b, dev), act=K.ACT_LRELU, gain=2 ** 0.5, alpha=0.5, out=out, ld_out=ldo, dtype=K.dt_code(dtype),
                     tile_hint=hint, **kw)
            return out, rgb_out
        monkeypatch.setenv("VT_PATCHW_WGS", str(wgs))
        got, got_rgb = run()
        monkeypatch.setenv("VT_PATCH_PIPE", "1")
        ref, ref_rgb = run()
        monkeypatch.delenv("VT_PATCH_PIPE")
        monkeypatch.delenv("VT_PATCHW_WGS")
        assert torch.equal(got, ref) and torch.equal(got_rgb, ref_rgb), (N, c0, c1, H, W, cout, hint, wgs)
        xq = torch.cat([xa] + ([xb] if c1 else []), dim=3).float().cpu().permute(0, 3, 1, 2).numpy()
        wq = wp.float().cpu().numpy().reshape(cout, 3, 3, cin).transpose(0, 3, 1, 2)
        want = O.leaky_relu(O.conv2d(xq, wq, b, 1, 1, 1), 0.2) * np.float32(2 ** 0.5) * 0.5
        if not rgb:
            want = want + 0.25 * r.float().cpu().permute(0, 3, 1, 2).numpy()
        assert rel_err(got.float().cpu().permute(0, 3, 1, 2).numpy(), want) < 8e-3


def test_conv_patch_weights_resident_equals_pipelined(dev, monkeypatch):
    """conv_patchw_kernel (csrc/conv_patch_resident.hpp): single-chunk layers (Cin = 64 bf16) on 256 x 64 tiles with all 9 taps
    of the weights resident in LDS and persistent workgroups that walk tiles with the next patch in flight.  Same K order as
    the pipelined / per-tap forms -> the same bits in the activation.  VT_PATCHW_WGS makes small convolutions walk several tiles per workgroup:
    3 (plain striding), 8 and 16 (tile ranges per XCD, one / two workgroups each), odd and even tile counts per workgroup (the
    two 4-wave groups of a workgroup take alternate tiles), ragged edges, a batch, bias + LeakyReLU, the fused ToRGB epilogue."""
    dtype = torch.bfloat16
    g = np.random.default_rng(41)
    for N, H, W, cout, wgs, rgb in ((1, 40, 50, 64, 3, False), (2, 33, 47, 64, 8, True), (1, 80, 100, 64, 16, False),
                                    (3, 17, 31, 64, 2, True)):
        cin = 64
        x = g.standard_normal((N, cin, H, W)).astype(np.float32)
        w = (g.standard_normal((cout, cin, 3, 3)) / math.sqrt(cin * 9)).astype(np.float32)
        b = g.standard_normal(cout).astype(np.float32)
        xa = K.nchw_to_nhwc(T(x, dev), dtype)
        wp = K.pack_conv_weight(T(w, dev), out_dtype=dtype)
        ldo = (cout + 7) // 8 * 8
        rgbw = K.pack_conv_weight(T((g.standard_normal((3, cout, 1, 1)) / 8).astype(np.float32), dev), out_dtype=dtype)
        skip = T(g.standard_normal((N, 3, H, W)).astype(np.float32), dev)
        rgbb = T(g.standard_normal(3).astype(np.float32), dev)

        def run():
            out = torch.zeros((N, H, W, ldo), dtype=dtype, device=dev)
            rgb_out = torch.zeros((N, 3, H, W), dtype=torch.float32, device=dev)
            kw = dict(rgb_weight=rgbw, rgb_bias=rgbb, rgb_resid=skip, rgb_out=rgb_out) if rgb else {}
            K.conv2d(src0=xa, c0=cin, ld0=cin, n=N, h=H, w=W, out_h=H, out_w=W, weight=wp, cout=cout, kh=3, kw=3, pad=1,
                     bias=T(b, dev), act=K.ACT_LRELU, gain=2 ** 0.5, alpha=0.5, out=out, ld_out=ldo, dtype=K.dt_code(dtype),
                     tile_hint=P + 256064, **kw)
            return out, rgb_out
        monkeypatch.setenv("VT_PATCHW_WGS", str(wgs))
        got, got_rgb = run()
        monkeypatch.setenv("VT_PATCH_PIPE", "1")
        ref, ref_rgb = run()
        monkeypatch.delenv("VT_PATCH_PIPE")
        monkeypatch.delenv("VT_PATCHW_WGS")
        # (the ToRGB sums run over 64 channels in ONE wave here, over two waves' halves through LDS there: same terms, another
        # association -- the image agrees to fp32 rounding, the activation bit for bit)
        assert torch.equal(got, ref), (N, H, W, cout, wgs)
        assert rel_err(got_rgb.cpu().numpy(), ref_rgb.cpu().numpy()) < 2e-6, (N, H, W, cout, wgs)
        wq = wp.float().cpu().numpy().reshape(cout, 3, 3, cin).transpose(0, 3, 1, 2)
        y = O.leaky_relu(O.conv2d(xa.float().cpu().permute(0, 3, 1, 2).numpy(), wq, b, 1, 1, 1), 0.2) * np.float32(2 ** 0.5) * 0.5
        assert rel_err(got.float().cpu().permute(0, 3, 1, 2)[:, :cout].numpy(), y) < 8e-3


def test_conv_f32x3(dev):
    """vt_conv_desc.dtype = VT_F32X3 (ABI 5): fp32 tensors, every product as three bf16 MFMAs (bf16 head / remainder split of
    both operands in the fragment registers).  Patch tiles, the direct-to-LDS 1-D kernel (stride 2) and the whole-K kernel
    (fragment stream) against the fp32 oracle at 3e-5 of max|y| -- and NOT bit-equal to the exact-fp32 instance, i.e. the
    f32x3 instance really ran; a conv whose kernel has no such instance (thin outputs) runs exact fp32."""
    import ctypes
    from vtoonify_amd import _lib
    g = np.random.default_rng(77)
    cases = [  # N, cin, H, W, cout, stride, dil, hint, stream, expect_x3
        (2, 64, 19, 37, 72, 1, 1, 0, False, True),                   # patch, auto plan
        (1, 96, 21, 35, 136, 1, 1, P + 256128, False, True),         # 256 x 128 patch tiles, 3 chunks
        (1, 64, 13, 18, 72, 2, 1, 0, False, True),                   # stride 2: 1-D direct-to-LDS kernel
        (1, 64, 19, 37, 72, 1, 1, P + 256064, False, True),          # 256 x 64 patch tiles (the per-tap form), 2 chunks
        (2, 256, 9, 11, 136, 1, 1, 0, True, True),                   # whole-K kernel (one round of 8 x 32 channels)
        (1, 256, 12, 10, 40, 1, 2, 4 * P, True, True),               # whole-K, dilated, forced by hint
        (1, 64, 9, 9, 3, 1, 1, 0, False, False),                     # thin outputs: exact fp32
    ]
    for N, cin, H, W, cout, stride, dil, hint, stream, expect_x3 in cases:
        x = g.standard_normal((N, cin, H, W)).astype(np.float32)
        w = (g.standard_normal((cout, cin, 3, 3)) / math.sqrt(cin * 9)).astype(np.float32)
        b = g.standard_normal(cout).astype(np.float32)
        pad = dil
        Ho, Wo = (H + 2 * pad - 2 * dil - 1) // stride + 1, (W + 2 * pad - 2 * dil - 1) // stride + 1
        xt = K.nchw_to_nhwc(T(x, dev), torch.float32)
        wp = K.pack_conv_weight(T(w, dev), out_dtype=torch.float32)
        wst = K.conv_weight_stream(wp) if stream else None
        ws = torch.zeros(8 << 20, dtype=torch.float32, device=dev)
        planar = cout <= 3

        def run(dt):
            if planar:
                out = torch.zeros((N, cout, Ho, Wo), dtype=torch.float32, device=dev)
                okw = dict(out=out, ld_out=0, out_layout=K.OUT_NCHW, out_dtype=K.VT_F32)
            else:
                out = torch.zeros((N, Ho, Wo, (cout + 7) // 8 * 8), dtype=torch.float32, device=dev)
                okw = dict(out=out, ld_out=out.shape[3], out_dtype=K.VT_F32)
            K.conv2d(src0=xt, c0=cin, ld0=cin, n=N, h=H, w=W, out_h=Ho, out_w=Wo, weight=wp, cout=cout, kh=3, kw=3,
                     stride=stride, pad=pad, dil=dil, bias=T(b, dev), act=K.ACT_LRELU, gain=2 ** 0.5, dtype=dt,
                     tile_hint=hint, splitk_ws=ws, weight_stream=wst, **okw)
            return out if planar else out.permute(0, 3, 1, 2)[:, :cout]
        y3, y1 = run(K.VT_F32X3), run(K.VT_F32)
        ref = O.leaky_relu(O.conv2d(x, w, b, stride, pad, dil), 0.2) * np.float32(2 ** 0.5)
        assert rel_err(y1.cpu().numpy(), ref) < F32_TOL
        assert rel_err(y3.cpu().numpy(), ref) < 3e-5, (cin, H, W, cout, stride, dil, hint)
        assert torch.equal(y3, y1) != expect_x3, (cin, H, W, cout, stride, dil, hint)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_transpose_by_parity(dev, dtype):
    """conv_transpose2d(3x3, stride 2, pad 0) with wide channels -- what the reference's StyledConv(upsample) calls through
    conv2d_gradfix.conv_transpose2d (model/stylegan/model.py:273-283) -- by output parity on the pipelined patch tiles
    (conv_patch_pipe.hpp, UP = 1: plan kind 1 for a TRANSPOSED descriptor; four accumulator sets, the nine weight taps as
    nine GEMM steps over the resident patch, 9 MACs per input pixel where the gather form visits 36 and skips 27) against
    the oracle and against the register-staged gather form; ragged quad grids, a batch, a channel count no tile divides."""
    import ctypes
    from vtoonify_amd import _lib
    unit = 64 if dtype == torch.bfloat16 else 32
    g = np.random.default_rng(5)
    t = F32_TOL if dtype == torch.float32 else 8e-3
    for N, cin, H, W, cout in ((2, 128, 9, 21, 72), (1, 4 * unit + 128, 16, 15, 136), (1, 128, 3, 33, 64)):
        x = g.standard_normal((N, cin, H, W)).astype(np.float32)
        w = (g.standard_normal((cin, cout, 3, 3)) / math.sqrt(cin * 9)).astype(np.float32)   # conv_transpose2d layout
        xt = K.nchw_to_nhwc(T(x, dev), dtype)
        wp = K.pack_conv_weight(T(w, dev), src_transposed=True, out_dtype=dtype)
        ldo = (cout + 7) // 8 * 8
        z = torch.zeros((N, 2 * H + 1, 2 * W + 1, ldo), dtype=dtype, device=dev)
        kw = dict(src0=xt, c0=cin, ld0=cin, n=N, h=H, w=W, out_h=2 * H + 1, out_w=2 * W + 1, weight=wp, cout=cout, kh=3, kw=3,
                  stride=2, pad=0, transposed=1, out=z, ld_out=ldo, dtype=K.dt_code(dtype))
        code = _lib.lib().vt_conv2d_tile(ctypes.byref(K.make_conv_desc(**kw)))
        assert code // 100000000 == 1 and code % 1000000 == 256064, code     # the parity-class patch tiles
        K.conv2d(**kw)
        xq = xt.float().cpu().permute(0, 3, 1, 2).numpy()
        wq = wp.float().cpu().numpy().reshape(cout, 3, 3, cin).transpose(3, 0, 1, 2)          # back to (cin, cout, kh, kw)
        zr = O.conv_transpose2d(xq, wq, 2)
        zg = z.float().cpu().permute(0, 3, 1, 2)[:, :cout].numpy()
        assert rel_err(zg, zr) < t, (N, cin, H, W, cout)
        # the same conv on the register-staged gather form (forced): same result to rounding
        z2 = torch.zeros_like(z)
        K.conv2d(**{**kw, "out": z2, "tile_hint": 1000000000})
        assert rel_err(z2.float().cpu().numpy(), z.float().cpu().numpy()) < t


def test_conv_batch_invariance(dev):
    """Tile / split-K choices depend on the per-image geometry only, so a frame convolved inside a
    batch is BIT-identical to the same frame alone (video path: s_w.repeat(B,1,1))."""
    g = np.random.default_rng(5)
    x = g.standard_normal((3, 128, 12, 20)).astype(np.float32)
    w = (g.standard_normal((64, 128, 3, 3)) / 30).astype(np.float32)
    wp = K.pack_conv_weight(T(w, dev), out_dtype=torch.bfloat16)
    ws = torch.zeros(8 << 20, dtype=torch.float32, device=dev)

    def run(xb):
        xt = K.nchw_to_nhwc(T(xb, dev), torch.bfloat16)
        out = torch.zeros((xb.shape[0], 12, 20, 64), dtype=torch.bfloat16, device=dev)
        K.conv2d(src0=xt, c0=128, ld0=128, n=xb.shape[0], h=12, w=20, out_h=12, out_w=20, weight=wp, cout=64,
                 kh=3, kw=3, pad=1, out=out, ld_out=64, dtype=K.VT_BF16, splitk_ws=ws)
        return out

    yb = run(x)
    for i in range(3):
        assert torch.equal(yb[i:i + 1], run(x[i:i + 1])), i


def test_conv_splitk_in_launch_equals_two_pass(dev, monkeypatch):
    """The last-arriving K-slice reduces in slice order: same bits as the separate reduce kernel,
    run after run (the arrival counters re-arm themselves)."""
    g = np.random.default_rng(9)
    x = g.standard_normal((1, 256, 8, 8)).astype(np.float32)
    w = (g.standard_normal((72, 256, 3, 3)) / 40).astype(np.float32)
    wp = K.pack_conv_weight(T(w, dev), out_dtype=torch.bfloat16)
    xt = K.nchw_to_nhwc(T(x, dev), torch.bfloat16)
    ws = torch.zeros(4 << 20, dtype=torch.float32, device=dev)

    def run(hint):
        out = torch.zeros((1, 8, 8, 72), dtype=torch.bfloat16, device=dev)
        K.conv2d(src0=xt, c0=256, ld0=256, n=1, h=8, w=8, out_h=8, out_w=8, weight=wp, cout=72, kh=3, kw=3, pad=1,
                 act=K.ACT_LRELU, out=out, ld_out=72, dtype=K.VT_BF16, splitk_ws=ws, tile_hint=hint)
        return out

    for hint in (2 * P + 4000000 + 64064, P + 4000000 + 128064):
        b = run(hint)                             # default: separate reduce kernel
        monkeypatch.setenv("VT_SPLITK_IN_LAUNCH", "1")
        a = run(hint)
        assert torch.equal(run(hint), a)          # counters re-armed, deterministic
        monkeypatch.delenv("VT_SPLITK_IN_LAUNCH")
        assert torch.equal(a, b), hint
        assert int(ws.view(torch.int32)[:4096].abs().max()) == 0   # ticket area left zero

    # a thin output (mask / ToRGB / fusion_skip shape, planar fp32): both forms, and the host query that names them
    import ctypes as C
    from vtoonify_amd import _lib
    lib = _lib.lib()
    w3 = (g.standard_normal((3, 256, 3, 3)) / 40).astype(np.float32)
    w3p = K.pack_conv_weight(T(w3, dev), out_dtype=torch.bfloat16)
    bias = T(g.standard_normal(3).astype(np.float32), dev)

    def thin():
        out = torch.zeros((1, 3, 8, 8), dtype=torch.float32, device=dev)
        d = K.make_conv_desc(src0=xt, c0=256, ld0=256, n=1, h=8, w=8, out_h=8, out_w=8, weight=w3p, cout=3, kh=3, kw=3,
                             pad=1, bias=bias, out=out, ld_out=0, out_layout=K.OUT_NCHW, out_dtype=K.VT_F32,
                             dtype=K.VT_BF16, splitk_ws=ws, tile_hint=128016)   # (no hint: the thin kernel, no split)
        mode = lib.vt_conv2d_splitk_mode(C.byref(d))
        assert (lib.vt_conv2d_tile(C.byref(d)) // 1000000) % 100 > 1     # really split
        _lib.check(lib.vt_conv2d(C.byref(d), K._stream(out)), "conv")
        return out, mode

    b, mode = thin()
    assert mode == 2                              # default: slices + reduce kernel
    monkeypatch.setenv("VT_SPLITK_IN_LAUNCH", "1")
    a, mode = thin()
    assert mode == 1 and torch.equal(thin()[0], a)
    monkeypatch.delenv("VT_SPLITK_IN_LAUNCH")
    assert torch.equal(a, b)
    assert int(ws.view(torch.int32)[:4096].abs().max()) == 0
    ref = torch.nn.functional.conv2d(xt.float().permute(0, 3, 1, 2).cpu(), w3p.float().view(3, 3, 3, 256).permute(0, 3, 1, 2).cpu(),
                                     bias.cpu(), padding=1)
    assert rel_err(a.cpu().numpy(), ref.numpy()) < 1e-5


def test_conv_concat_prologue_transposed(dev):
    g = np.random.default_rng(0)
    N, C0, C1, H, W, Co = 1, 16, 24, 7, 6, 40
    a = g.standard_normal((N, C0, H, W)).astype(np.float32)
    b = g.standard_normal((N, C1, H, W)).astype(np.float32)
    w = (g.standard_normal((Co, C0 + C1, 3, 3)) / 15).astype(np.float32)
    at = K.nchw_to_nhwc(T(a, dev), torch.float32)
    bt = K.nchw_to_nhwc(T(b, dev), torch.float32, ld_out=32)  # wider pixel stride than channels
    out = torch.zeros((N, H, W, Co), device=dev)
    K.conv2d(src0=at, c0=C0, ld0=C0, src1=bt, c1=C1, ld1=32, n=N, h=H, w=W, out_h=H, out_w=W,
             weight=K.pack_conv_weight(T(w, dev)), cout=Co, kh=3, kw=3, pad=1, out=out, ld_out=Co, dtype=0)
    assert rel_err(out.cpu().permute(0, 3, 1, 2).numpy(), O.conv2d(np.concatenate([a, b], 1), w, None, 1, 1, 1)) < F32_TOL
    sc = g.standard_normal((N, C0)).astype(np.float32)
    sh = g.standard_normal((N, C0)).astype(np.float32)
    w2 = (g.standard_normal((Co, C0, 3, 3)) / 10).astype(np.float32)
    out = torch.zeros((N, H, W, Co), device=dev)
    K.conv2d(src0=at, c0=C0, ld0=C0, n=N, h=H, w=W, out_h=H, out_w=W, weight=K.pack_conv_weight(T(w2, dev)),
             cout=Co, kh=3, kw=3, pad=1, in_scale=T(sc, dev), in_shift=T(sh, dev), out=out, ld_out=Co, dtype=0)
    ref = O.conv2d(a * sc.reshape(N, C0, 1, 1) + sh.reshape(N, C0, 1, 1), w2, None, 1, 1, 1)
    assert rel_err(out.cpu().permute(0, 3, 1, 2).numpy(), ref) < F32_TOL


def test_conv2d_gradfix_surface(dev):
    g = np.random.default_rng(4)
    x = g.standard_normal((2, 6, 9, 7)).astype(np.float32)
    w = (g.standard_normal((8, 3, 3, 3)) / 5).astype(np.float32)
    b = g.standard_normal(8).astype(np.float32)
    y = op.conv2d_gradfix.conv2d(T(x, dev), T(w, dev), T(b, dev), stride=1, padding=1, groups=2)
    ref = np.concatenate([O.conv2d(x[:, :3], w[:4], b[:4], 1, 1, 1), O.conv2d(x[:, 3:], w[4:], b[4:], 1, 1, 1)], 1)
    assert rel_err(y.cpu().numpy(), ref) < F32_TOL
    wt = (g.standard_normal((6, 5, 3, 3)) / 5).astype(np.float32)
    yt = op.conv2d_gradfix.conv_transpose2d(T(x, dev), T(wt, dev), stride=2, padding=0)
    assert rel_err(yt.cpu().numpy(), O.conv_transpose2d(x, wt, 2)) < F32_TOL


def _gradfix_case(dev, transposed, N, Ci, H, W, Co, k, s, p, d, groups=1, opad=0, seed=0):
    """conv2d_gradfix autograd (first and second order) against torch's own autograd of F.conv2d /
    F.conv_transpose2d on the CPU -- what the reference's Conv2d / Conv2dGradWeight Functions
    (op/conv2d_gradfix.py:134-223) delegate to."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(seed)
    x0 = torch.randn(N, Ci, H, W, generator=g)
    w0 = torch.randn((Ci, Co // groups, k, k) if transposed else (Co, Ci // groups, k, k), generator=g) / 3
    b0 = torch.randn(Co, generator=g)
    x, w, b = [t.clone().to(dev).requires_grad_(True) for t in (x0, w0, b0)]
    xr, wr, br = [t.clone().requires_grad_(True) for t in (x0, w0, b0)]
    if transposed:
        y = op.conv2d_gradfix.conv_transpose2d(x, w, b, stride=s, padding=p, output_padding=opad, groups=groups, dilation=d)
        yr = F.conv_transpose2d(xr, wr, br, stride=s, padding=p, output_padding=opad, groups=groups, dilation=d)
    else:
        y = op.conv2d_gradfix.conv2d(x, w, b, stride=s, padding=p, dilation=d, groups=groups)
        yr = F.conv2d(xr, wr, br, stride=s, padding=p, dilation=d, groups=groups)
    go = torch.randn(yr.shape, generator=g)
    gx, gw, gb = torch.autograd.grad(y, (x, w, b), go.to(dev), create_graph=True)
    rx, rw, rb = torch.autograd.grad(yr, (xr, wr, br), go, create_graph=True)
    u, v = torch.randn(rx.shape, generator=g), torch.randn(rw.shape, generator=g)
    hx, hw = torch.autograd.grad((gx * u.to(dev)).sum() + (gw * v.to(dev)).sum(), (x, w))   # R1-style second order
    qx, qw = torch.autograd.grad((rx * u).sum() + (rw * v).sum(), (xr, wr))
    pairs = ((y, yr), (gx, rx), (gw, rw), (gb, rb), (hx, qx), (hw, qw))
    return max(rel_err(a.detach().cpu().numpy(), r.detach().numpy()) for a, r in pairs)


def test_conv2d_gradfix_autograd(dev):
    assert _gradfix_case(dev, False, 2, 8, 9, 7, 6, 3, 1, 1, 1) < F32_TOL
    assert _gradfix_case(dev, False, 2, 4, 10, 9, 8, 3, 2, 1, 1, seed=1) < F32_TOL          # stride 2, odd sizes
    assert _gradfix_case(dev, False, 1, 8, 9, 9, 4, 3, 1, 2, 2, seed=2) < F32_TOL           # dilation 2
    assert _gradfix_case(dev, True, 2, 6, 5, 4, 8, 3, 2, 0, 1, seed=3) < F32_TOL            # StyledConv up-sampling form
    assert _gradfix_case(dev, True, 2, 6, 5, 6, 8, 3, 2, 1, 1, opad=1, seed=4) < F32_TOL
    assert _gradfix_case(dev, False, 2, 8, 6, 6, 6, 3, 1, 1, 1, groups=2, seed=5) < F32_TOL  # ModulatedConv2d: groups = batch
    assert _gradfix_case(dev, True, 2, 8, 5, 5, 6, 3, 2, 0, 1, groups=2, seed=6) < F32_TOL
    # no_weight_gradients() (util.py:76: path-length regulariser) suppresses grad_weight only
    x = torch.randn(1, 8, 6, 6).to(dev).requires_grad_(True)
    w = torch.randn(4, 8, 3, 3).to(dev).requires_grad_(True)
    with op.conv2d_gradfix.no_weight_gradients():
        gx, gw = torch.autograd.grad(op.conv2d_gradfix.conv2d(x, w, padding=1).sum(), (x, w), allow_unused=True)
    assert gx is not None and gw is None


def test_conv2d_gradfix_autograd_wide_channels(dev):
    """The same with >= 32 output channels per group: NHWC out of the fast epilogues + the tiled vt_nhwc_to_nchw, and
    conv_transpose2d(stride 1) -- every stride-1 grad_input -- as a convolution with the turned kernel."""
    assert _gradfix_case(dev, False, 2, 32, 9, 7, 40, 3, 1, 1, 1, seed=10) < F32_TOL
    assert _gradfix_case(dev, False, 1, 16, 10, 9, 32, 3, 2, 1, 1, seed=11) < F32_TOL          # stride 2
    assert _gradfix_case(dev, False, 1, 32, 9, 9, 32, 3, 1, 2, 2, seed=12) < F32_TOL           # dilation 2
    assert _gradfix_case(dev, True, 2, 32, 6, 5, 32, 3, 1, 1, 1, seed=13) < F32_TOL            # transposed, stride 1
    assert _gradfix_case(dev, True, 1, 16, 5, 4, 32, 3, 2, 0, 1, seed=14) < F32_TOL            # transposed, stride 2
    assert _gradfix_case(dev, False, 2, 16, 6, 6, 64, 3, 1, 1, 1, groups=2, seed=15) < F32_TOL  # groups: 32 channels each
    assert _gradfix_case(dev, False, 1, 32, 8, 8, 32, 1, 1, 0, 1, seed=16) < F32_TOL           # 1x1
    # bf16 tensors: forward and first-order gradients against the same graph on the bf16-rounded operands in fp32
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(17)
    x0 = torch.randn(2, 32, 8, 7, generator=g).bfloat16()
    w0 = (torch.randn(40, 32, 3, 3, generator=g) / 17).bfloat16()
    go = torch.randn(2, 40, 8, 7, generator=g).bfloat16()
    x, w = x0.to(dev).requires_grad_(True), w0.to(dev).requires_grad_(True)
    y = op.conv2d_gradfix.conv2d(x, w, padding=1)
    gx, gw = torch.autograd.grad(y, (x, w), go.to(dev))
    xr, wr = x0.float().requires_grad_(True), w0.float().requires_grad_(True)
    yr = F.conv2d(xr, wr, padding=1)
    rx, rw = torch.autograd.grad(yr, (xr, wr), go.float())
    assert y.dtype == torch.bfloat16 and gx.dtype == torch.bfloat16 and gw.dtype == torch.bfloat16
    for a, r in ((y, yr), (gx, rx), (gw, rw)):
        assert rel_err(a.detach().float().cpu().numpy(), r.detach().numpy()) < 1.2e-2


# ---------------------------------------------------------------- style ops / norm / glue
def test_linear_pixelnorm(dev):
    g = np.random.default_rng(0)
    x = g.standard_normal((5, 24)).astype(np.float32)
    W = g.standard_normal((40, 24)).astype(np.float32)
    b = g.standard_normal(40).astype(np.float32)
    y = K.linear(T(x, dev), T(W, dev), T(b, dev), w_scale=0.3, b_scale=0.01, act=K.ACT_LRELU, slope=0.2, gain=2 ** 0.5)
    assert rel_err(y.cpu().numpy(), O.fused_leaky_relu(x @ (W * np.float32(0.3)).T, b * np.float32(0.01))) < F32_TOL
    assert rel_err(K.pixel_norm(T(x, dev)).cpu().numpy(), O.pixel_norm(x)) < 1e-6
    d, _ = load_golden("modules.npz")
    y = K.linear(T(d["el_act__x"], dev), T(d["el_act.weight"], dev), T(d["el_act.bias"], dev),
                 w_scale=(1 / math.sqrt(24)) * 0.01, b_scale=0.01, act=K.ACT_LRELU, slope=0.2, gain=2 ** 0.5)
    assert rel_err(y.cpu().numpy(), d["el_act__y"]) < F32_TOL


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_styled_conv_golden(dev, dtype):
    """ModulatedConv2d weight path + polyphase up-sampling conv vs the REAL reference's
    StyledConv outputs (conv_transpose2d + blur + bias + lrelu), per-sample styles."""
    d, _ = load_golden("modules.npz")
    for name, up in (("sc_up", True), ("sc_same", False)):
        w = d[name + ".conv.weight"][0]
        cout, cin = w.shape[:2]
        xs, ss, ref = d[name + "__x"], d[name + "__s"], d[name + "__y"]
        N, _, H, W = xs.shape
        outs = []
        for bi in range(N):
            s = K.linear(T(ss[bi:bi + 1].copy(), dev), T(d[name + ".conv.modulation.weight"], dev),
                         T(d[name + ".conv.modulation.bias"], dev), w_scale=1 / math.sqrt(32))
            fir = T(d[name + ".conv.blur.kernel"], dev) if up else None
            wp = K.modulate_weight(T(w.copy(), dev), s[0].contiguous(), 1 / math.sqrt(cin * 9), True, fir=fir,
                                   out_dtype=dtype)
            xt = K.nchw_to_nhwc(T(xs[bi:bi + 1].copy(), dev), dtype)
            f = 2 if up else 1
            out = torch.zeros((1, H * f, W * f, cout), dtype=dtype, device=dev)
            K.conv2d(src0=xt, c0=cin, ld0=cin, n=1, h=H, w=W, out_h=H, out_w=W, weight=wp, cout=cout, kh=3, kw=3,
                     pad=1, phases=4 if up else 1, out=out, ld_out=cout, dtype=K.dt_code(dtype),
                     bias=T(d[name + ".activate.bias"], dev), act=K.ACT_LRELU, gain=2 ** 0.5)
            outs.append(out.float().cpu().permute(0, 3, 1, 2).numpy())
        assert rel_err(np.concatenate(outs, 0), ref) < tol(dtype), name


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_instnorm_adain_fusion_pack(dev, dtype):
    g = np.random.default_rng(1)
    N, Cc, H, W = 2, 16, 37, 29  # > 1 statistics chunk
    x = (g.standard_normal((N, Cc, H, W)) * 2 + 1).astype(np.float32)
    o = g.standard_normal((N, Cc, H, W)).astype(np.float32)
    xt, ot = K.nchw_to_nhwc(T(x, dev), dtype), K.nchw_to_nhwc(T(o, dev), dtype)
    xq = xt.float().cpu().permute(0, 3, 1, 2).numpy()
    oq = ot.float().cpu().permute(0, 3, 1, 2).numpy()
    for use_other in (False, True):
        ct = Cc * (2 if use_other else 1)
        gb = g.standard_normal((N, 2 * ct)).astype(np.float32)
        scale = torch.zeros((N, ct), device=dev)
        shift = torch.zeros((N, ct), device=dev)
        ws = torch.zeros(K.instnorm_ws_bytes(N, H * W, ct), dtype=torch.uint8, device=dev)
        K.instnorm_stats(scale, shift, xt, Cc, N, H * W, Cc, ws, K.dt_code(dtype), other=ot if use_other else None,
                         ld_other=Cc, style_gb=T(gb, dev), ld_gb=2 * ct)
        out = torch.zeros((N, H, W, ct), dtype=dtype, device=dev)
        K.affine_apply(out, ct, xt, Cc, scale, shift, N, H * W, Cc, K.dt_code(dtype), other=ot if use_other else None,
                       ld_other=Cc)
        full = np.concatenate([xq, np.abs(xq - oq)], 1) if use_other else xq
        ref = gb[:, :ct].reshape(N, ct, 1, 1) * O.instance_norm(full) + gb[:, ct:].reshape(N, ct, 1, 1)
        assert rel_err(out.float().cpu().permute(0, 3, 1, 2).numpy(), ref) < (F32_TOL if dtype == torch.float32 else 8e-3)
    mask = g.random((N, H, W)).astype(np.float32)
    skip = g.standard_normal((N, 3, H, W)).astype(np.float32)
    out = torch.full((N, H, W, Cc + 8), 7.0, dtype=dtype, device=dev)
    K.fusion_pack(out, Cc + 8, xt, Cc, T(mask, dev), T(skip, dev), N, H * W, Cc, K.dt_code(dtype))
    o_ = out.float().cpu().numpy()
    t = 1e-6 if dtype == torch.float32 else 8e-3
    assert rel_err(o_[..., :3], skip.transpose(0, 2, 3, 1)) < t and np.abs(o_[..., 3:8]).max() == 0
    assert rel_err(o_[..., 8:], xq.transpose(0, 2, 3, 1) * mask[..., None]) < t


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_instnorm_plane_one_launch(dev, dtype):
    """vt_instnorm_plane: AdaIN of a small plane, statistics included, one launch (dualstylegan.py:6-21 -- biased variance,
    eps 1e-5, style affine).  Every registers-per-thread instance (1 / 4 / 8 / 16 pixels per thread), ragged plane sizes,
    in-place operation, no style, an image alone == the same image inside a batch (bit-exact), too-large planes refused."""
    from vtoonify_amd import _lib
    lib = _lib.lib()
    g = np.random.default_rng(31)
    code = K.dt_code(dtype)
    for N, Cc, H, W in ((2, 16, 16, 16), (2, 24, 32, 32), (3, 16, 37, 29), (1, 8, 45, 50), (1, 16, 64, 64),
                        (2, 64, 32, 32), (1, 32, 16, 15), (2, 32, 40, 33)):   # 64-byte rows per workgroup (VPW 4, 2)
        hw = H * W
        x = (g.standard_normal((N, Cc, H, W)) * 2 + 1).astype(np.float32)
        gb = g.standard_normal((N, 2 * Cc)).astype(np.float32)
        xt = K.nchw_to_nhwc(T(x, dev), dtype)
        xq = xt.float().cpu().permute(0, 3, 1, 2).numpy()
        gbt = T(gb, dev)
        out = torch.zeros_like(xt)
        assert lib.vt_instnorm_plane(out.data_ptr(), Cc, xt.data_ptr(), Cc, None, 0, N, hw, Cc, gbt.data_ptr(), 2 * Cc, code,
                                     K._stream(xt)) == 0, lib.vt_last_error()
        ref = gb[:, :Cc].reshape(N, Cc, 1, 1) * O.instance_norm(xq) + gb[:, Cc:].reshape(N, Cc, 1, 1)
        assert rel_err(out.float().cpu().permute(0, 3, 1, 2).numpy(), ref) < (F32_TOL if dtype == torch.float32 else 8e-3)
        inpl = xt.clone()                                       # out aliases x
        assert lib.vt_instnorm_plane(inpl.data_ptr(), Cc, inpl.data_ptr(), Cc, None, 0, N, hw, Cc, gbt.data_ptr(), 2 * Cc, code,
                                     K._stream(xt)) == 0
        assert torch.equal(inpl, out)
        one = torch.zeros_like(xt[-1:])                         # the last image alone
        assert lib.vt_instnorm_plane(one.data_ptr(), Cc, xt[-1:].contiguous().data_ptr(), Cc, None, 0, 1, hw, Cc,
                                     gbt[-1:].contiguous().data_ptr(), 2 * Cc, code, K._stream(xt)) == 0
        assert torch.equal(one[0], out[-1])
        plain = torch.zeros_like(xt)                            # no style: InstanceNorm2d alone
        assert lib.vt_instnorm_plane(plain.data_ptr(), Cc, xt.data_ptr(), Cc, None, 0, N, hw, Cc, None, 0, code,
                                     K._stream(xt)) == 0
        assert rel_err(plain.float().cpu().permute(0, 3, 1, 2).numpy(), O.instance_norm(xq)) < \
            (F32_TOL if dtype == torch.float32 else 8e-3)
        # Fusion.forward's norm (vtoonify.py:125): AdaIN of cat[x, |x - other|] into 2C channels, against the oracle and,
        # bit for bit up to the statistics' rounding, against the two-launch path (vt_instnorm_stats + vt_affine_apply)
        o = g.standard_normal((N, Cc, H, W)).astype(np.float32)
        ot = K.nchw_to_nhwc(T(o, dev), dtype)
        oq = ot.float().cpu().permute(0, 3, 1, 2).numpy()
        gb2 = g.standard_normal((N, 4 * Cc)).astype(np.float32)
        cat = torch.zeros((N, H, W, 2 * Cc), dtype=dtype, device=dev)
        assert lib.vt_instnorm_plane(cat.data_ptr(), 2 * Cc, xt.data_ptr(), Cc, ot.data_ptr(), Cc, N, hw, Cc,
                                     T(gb2, dev).data_ptr(), 4 * Cc, code, K._stream(xt)) == 0, lib.vt_last_error()
        full = np.concatenate([xq, np.abs(xq - oq)], 1)
        ref = gb2[:, :2 * Cc].reshape(N, 2 * Cc, 1, 1) * O.instance_norm(full) + gb2[:, 2 * Cc:].reshape(N, 2 * Cc, 1, 1)
        assert rel_err(cat.float().cpu().permute(0, 3, 1, 2).numpy(), ref) < (F32_TOL if dtype == torch.float32 else 8e-3)
        assert lib.vt_instnorm_plane(xt.data_ptr(), 2 * Cc, xt.data_ptr(), Cc, ot.data_ptr(), Cc, N, hw, Cc, None, 0, code,
                                     K._stream(xt)) == 1   # VT_ERR_ARG: the cat form cannot run in place
    big = torch.zeros((1, 65, 64, 8), dtype=dtype, device=dev)
    assert lib.vt_instnorm_plane(big.data_ptr(), 8, big.data_ptr(), 8, None, 0, 1, 65 * 64, 8, None, 0, code, K._stream(big)) == 2   # VT_ERR_UNSUPPORTED


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_whole_k_kernel(dev, dtype):
    """conv_fullk_kernel (vt_conv_desc.weight_stream): K split across the 8 wavefronts of a workgroup and
    summed through LDS -- no split-K slabs.  Covers one and two rounds of 8 channel chunks, two concat
    sources, dilation 1/2/4 (strided tiles), odd image sizes (partial tiles), a cout tail inside the last
    32-channel tile, batch, the residual / activation epilogue and planar output; compared with the oracle on
    the same rounded operands, and with the slab path on the same inputs."""
    import ctypes
    from vtoonify_amd import _lib
    g = np.random.default_rng(77)
    t = F32_TOL if dtype == torch.float32 else 8e-3
    unit = 256 if dtype == torch.float32 else 512          # 8 wavefronts x one 128-byte row of channels
    L = K.ACT_LRELU
    cases = [  # (N, c0, c1, H, W, Cout, dil, act, resid, planar)
        (2, unit, 0, 9, 11, 136, 1, L, True, False),            # cout >= 128: chosen without a hint
        (1, unit, 0, 9, 11, 40, 2, 0, True, False),
        (2, unit, 0, 5, 13, 64, 4, L, False, False),
        (1, 2 * unit, 0, 7, 9, 32, 1, L, True, False),          # two rounds
        (1, unit, unit, 6, 10, 40, 1, 0, False, False),         # two sources (torch.cat of Fusion)
        (1, unit, 0, 10, 9, 8, 1, K.ACT_RELU_TANH, False, True),  # planar fp32 output
    ]
    for N, c0, c1, H, W, Cout, dil, act, resid, planar in cases:
        cin = c0 + c1
        x = g.standard_normal((N, cin, H, W)).astype(np.float32)
        w = (g.standard_normal((Cout, cin, 3, 3)) / math.sqrt(cin * 9)).astype(np.float32)
        b = g.standard_normal(Cout).astype(np.float32)
        xt = K.nchw_to_nhwc(T(x, dev), dtype)
        wp = K.pack_conv_weight(T(w, dev), out_dtype=dtype)
        wst = K.conv_weight_stream(wp)
        assert wst is not None
        xq = xt.float().cpu().permute(0, 3, 1, 2).numpy()
        wq = wp.float().cpu().numpy().reshape(Cout, 3, 3, cin).transpose(0, 3, 1, 2)
        ref = O.conv2d(xq, wq, b, 1, dil, dil)
        gain = 1.0
        if act == L:
            ref, gain = O.leaky_relu(ref, 0.2) * np.float32(2 ** 0.5), 2 ** 0.5
        elif act == K.ACT_RELU_TANH:
            ref = np.tanh(np.maximum(ref, 0))
        if c1:   # the second source lives in its own tensor with a larger pixel stride
            x0 = xt[..., :c0].contiguous()
            x1 = torch.zeros((N, H, W, c1 + 16), dtype=dtype, device=dev)
            x1[..., :c1] = xt[..., c0:]
            srcs = dict(src0=x0, c0=c0, ld0=c0, src1=x1, c1=c1, ld1=c1 + 16)
        else:
            srcs = dict(src0=xt, c0=c0, ld0=c0)
        common = dict(n=N, h=H, w=W, out_h=H, out_w=W, weight=wp, cout=Cout, kh=3, kw=3, pad=dil, dil=dil,
                      bias=T(b, dev), act=act, gain=gain, dtype=K.dt_code(dtype), alpha=0.5 if resid else 1.0,
                      beta=0.25 if resid else 0.0, **srcs)
        outs = []
        for stream in (wst, None):
            if planar:
                out = torch.zeros((N, Cout, H, W), dtype=torch.float32, device=dev)
                kw = dict(out=out, ld_out=0, out_layout=K.OUT_NCHW, out_dtype=K.VT_F32)
                r = None
            else:
                out = torch.zeros((N, H, W, Cout), dtype=dtype, device=dev)
                r = None
                if resid:
                    rn = np.random.default_rng(5).standard_normal((N, Cout, H, W)).astype(np.float32)
                    r = K.nchw_to_nhwc(T(rn, dev), dtype, ld_out=Cout)
                kw = dict(out=out, ld_out=Cout, resid=r, ld_res=Cout)
            hint = 4 * P if (stream is not None and Cout < 128) else 0   # small cout: forced by hint
            d = K.make_conv_desc(weight_stream=stream, tile_hint=hint, **kw, **common)
            code = _lib.lib().vt_conv2d_tile(ctypes.byref(d))
            assert (code // 100000000 == 4) == (stream is not None), code
            K.conv2d(weight_stream=stream, tile_hint=hint, **kw, **common)
            outs.append(out.float().cpu().numpy() if planar else out.float().cpu().permute(0, 3, 1, 2).numpy())
        want = ref
        if resid and not planar:
            want = ref * 0.5 + 0.25 * r.float().cpu().permute(0, 3, 1, 2).numpy()
        assert rel_err(outs[0], want) < t, (N, c0, c1, H, W, Cout, dil)
        assert rel_err(outs[0], outs[1]) < t, "whole-K kernel vs the slab path"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_whole_k_adain_chain(dev, dtype):
    """AdaResBlock on the whole-K kernel (model/dualstylegan.py:38-45): conv A writes its output AND per-tile
    {mean, M2} records (vt_conv_desc.tile_stats); conv B merges them, applies AdaIN to its input patch in LDS
    and convolves -- no statistics / normalisation launches in between.  Checked: the records against numpy on
    the stored tensor, and conv B against conv(AdaIN(stored A)) from the oracle, for dilation 1/2/4 producers
    and consumers, odd sizes, batch 2 with per-sample gamma/beta and with a shared row."""
    g = np.random.default_rng(91)
    t = 1e-4 if dtype == torch.float32 else 1.2e-2
    unit = 256 if dtype == torch.float32 else 512
    L = K.ACT_LRELU
    for N, H, W, dA, dB, shared in [(2, 9, 11, 1, 2, False), (1, 13, 6, 4, 1, True), (1, 7, 10, 2, 4, True)]:
        C = unit
        x = g.standard_normal((N, C, H, W)).astype(np.float32)
        wA = (g.standard_normal((C, C, 3, 3)) / math.sqrt(C * 9)).astype(np.float32)
        wB = (g.standard_normal((40, C, 3, 3)) / math.sqrt(C * 9)).astype(np.float32)
        bA = g.standard_normal(C).astype(np.float32)
        gb = (1.0 + 0.3 * g.standard_normal((1 if shared else N, 2 * C))).astype(np.float32)
        xt = K.nchw_to_nhwc(T(x, dev), dtype)
        wpA, wpB = K.pack_conv_weight(T(wA, dev), out_dtype=dtype), K.pack_conv_weight(T(wB, dev), out_dtype=dtype)
        ts = torch.zeros(K.conv_tile_stats_bytes(N, H, W, dA, C) // 4, dtype=torch.float32, device=dev)
        yA = torch.zeros((N, H, W, C), dtype=dtype, device=dev)
        K.conv2d(src0=xt, c0=C, ld0=C, n=N, h=H, w=W, out_h=H, out_w=W, weight=wpA, weight_stream=K.conv_weight_stream(wpA),
                 cout=C, kh=3, kw=3, pad=dA, dil=dA, bias=T(bA, dev), act=L, gain=2 ** 0.5, out=yA, ld_out=C,
                 dtype=K.dt_code(dtype), tile_stats=ts)
        a = yA.float().cpu().permute(0, 3, 1, 2).numpy()                      # the tensor as stored
        # ---- records: tile (phase_y, phase_x, tile_y, tile_x) of dilation dA, {mean, M2} per channel
        ty, tx = -(-(-(-H // dA)) // 8), -(-(-(-W // dA)) // 8)
        nt = dA * dA * ty * tx
        rec = ts.cpu().numpy()[:N * nt * C * 2].reshape(N, nt, C, 2)
        cnt = ts.cpu().numpy()[N * nt * C * 2:].reshape(N, nt)
        for n in range(N):
            ti = 0
            for fy in range(dA):
                for fx in range(dA):
                    for iy in range(ty):
                        for ix in range(tx):
                            blk = a[n, :, fy + iy * 8 * dA::dA, fx + ix * 8 * dA::dA][:, :8, :8].reshape(C, -1)
                            assert cnt[n, ti] == blk.shape[1]
                            if blk.shape[1]:
                                m = blk.mean(1)
                                assert np.abs(rec[n, ti, :, 0] - m).max() < 1e-5 * max(1.0, np.abs(m).max())
                                m2 = ((blk - m[:, None]) ** 2).sum(1)
                                assert np.abs(rec[n, ti, :, 1] - m2).max() < 1e-4 * max(1.0, m2.max())
                            ti += 1
        # ---- conv B on AdaIN(a): reference from the stored tensor (rounded to `dtype` like nrm_res was)
        mean = a.mean((2, 3), keepdims=True)
        var = a.var((2, 3), keepdims=True)
        gam = np.broadcast_to(gb[:, :C, None, None], (N, C, 1, 1))
        bet = np.broadcast_to(gb[:, C:, None, None], (N, C, 1, 1))
        an = (a - mean) / np.sqrt(var + 1e-5) * gam + bet
        an = torch.from_numpy(an.astype(np.float32)).to(dtype).float().numpy()
        wq = wpB.float().cpu().numpy().reshape(40, 3, 3, C).transpose(0, 3, 1, 2)
        ref = O.conv2d(an, wq, None, 1, dB, dB)
        yB = torch.zeros((N, H, W, 40), dtype=dtype, device=dev)
        K.conv2d(src0=yA, c0=C, ld0=C, n=N, h=H, w=W, out_h=H, out_w=W, weight=wpB, weight_stream=K.conv_weight_stream(wpB),
                 cout=40, kh=3, kw=3, pad=dB, dil=dB, out=yB, ld_out=40, dtype=K.dt_code(dtype), tile_hint=4 * P,
                 in_tile_stats=ts, in_stats_dil=dA, in_gb=T(gb, dev), in_ld_gb=0 if shared else 2 * C)
        assert rel_err(yB.float().cpu().permute(0, 3, 1, 2).numpy(), ref) < t, (N, H, W, dA, dB)
    # both ends must be whole-K plans
    with pytest.raises(Exception, match="whole-K"):
        K.conv2d(src0=xt, c0=C, ld0=C, n=N, h=H, w=W, out_h=H, out_w=W, weight=wpA, cout=C, kh=3, kw=3, pad=1,
                 out=yA, ld_out=C, dtype=K.dt_code(dtype), tile_stats=ts)


@pytest.mark.parametrize("dtype", [torch.bfloat16])   # bf16 only: an fp32 512-channel conv is two rounds (conv_fullk_kernel)
def test_conv_weight_stationary_equals_whole_k(dev, dtype, monkeypatch):
    """conv_fullkw_kernel (conv_fullkw.hpp): the whole-K conv with the weights resident in registers and G tiles of one
    image streaming through a workgroup -- the batch form of the trunk convs (model/vtoonify.py:92-104,235-239 at
    style_transfer.py:35's --batch_size 4).  It must be BIT-IDENTICAL to conv_fullk_kernel for every G: outputs, the
    {mean, M2} tile records and tile pixel counts (vt_conv_desc.tile_stats), and the AdaIN-prologue consumer
    (in_tile_stats); dilation 1/2/4, partial tiles, a ragged last group (tiles % G != 0), cout tail, two sources,
    residual / d_s epilogue, planar output, batch with per-sample gamma/beta."""
    import ctypes
    from vtoonify_amd import _lib
    g = np.random.default_rng(123)
    unit = 256 if dtype == torch.float32 else 512          # single round: 8 wavefronts x one 128-byte row of channels
    L = K.ACT_LRELU
    P4 = 4 * P
    cases = [  # (N, c0, c1, H, W, Cout, dil, act, resid, planar, stats)
        (2, unit, 0, 17, 19, 136, 1, L, True, False, True),        # 3x3 tiles, partial edges, cout tail
        (3, unit, 0, 9, 11, 64, 2, L, False, False, True),         # 4 phases x (1x1) tiles
        (1, unit, 0, 13, 21, 40, 4, 0, True, False, True),         # 16 phases
        (2, unit // 2, unit // 2, 10, 17, 32, 1, L, False, False, False),   # two sources (torch.cat of Fusion)
        (1, unit, 0, 12, 9, 8, 1, K.ACT_RELU_TANH, False, True, False),      # planar fp32 output
    ]
    for N, c0, c1, H, W, Cout, dil, act, resid, planar, stats in cases:
        cin = c0 + c1
        x = g.standard_normal((N, cin, H, W)).astype(np.float32)
        w = (g.standard_normal((Cout, cin, 3, 3)) / math.sqrt(cin * 9)).astype(np.float32)
        b = g.standard_normal(Cout).astype(np.float32)
        xt = K.nchw_to_nhwc(T(x, dev), dtype)
        wp = K.pack_conv_weight(T(w, dev), out_dtype=dtype)
        wst = K.conv_weight_stream(wp)
        if c1:
            x0 = xt[..., :c0].contiguous()
            x1 = torch.zeros((N, H, W, c1 + 16), dtype=dtype, device=dev)
            x1[..., :c1] = xt[..., c0:]
            srcs = dict(src0=x0, c0=c0, ld0=c0, src1=x1, c1=c1, ld1=c1 + 16)
        else:
            srcs = dict(src0=xt, c0=c0, ld0=c0)
        ds = T(np.array([0.7], np.float32), dev)
        r = None
        if resid:
            rn = g.standard_normal((N, Cout, H, W)).astype(np.float32)
            r = K.nchw_to_nhwc(T(rn, dev), dtype, ld_out=Cout)
        common = dict(n=N, h=H, w=W, out_h=H, out_w=W, weight=wp, weight_stream=wst, cout=Cout, kh=3, kw=3, pad=dil,
                      dil=dil, bias=T(b, dev), act=act, gain=2 ** 0.5 if act == L else 1.0, dtype=K.dt_code(dtype),
                      alpha_dev=ds if resid else None, beta=1.0 if resid else 0.0, tile_hint=P4, **srcs)
        res = {}
        for G in (0, 1, 2, 4, 8):          # 0 = conv_fullk_kernel
            monkeypatch.setenv("VT_FULLKW", "0" if G == 0 else "1")
            monkeypatch.setenv("VT_FULLKW_MIN_G", "1")
            monkeypatch.setenv("VT_FULLKW_G", str(max(G, 1)))
            if planar:
                out = torch.zeros((N, Cout, H, W), dtype=torch.float32, device=dev)
                kw = dict(out=out, ld_out=0, out_layout=K.OUT_NCHW, out_dtype=K.VT_F32)
            else:
                out = torch.zeros((N, H, W, Cout), dtype=dtype, device=dev)
                kw = dict(out=out, ld_out=Cout, resid=r, ld_res=Cout)
            ts = None
            if stats:
                ts = torch.full((K.conv_tile_stats_bytes(N, H, W, dil, Cout) // 4,), -7.0, dtype=torch.float32, device=dev)
            d = K.make_conv_desc(tile_stats=ts, **kw, **common)
            code = _lib.lib().vt_conv2d_tile(ctypes.byref(d))
            # (a planar fp32 output stays on conv_fullk_kernel: the weight-stationary form has the lean NHWC epilogue only)
            assert code // 100000000 == (4 if (G == 0 or planar) else 8), (code, G)
            K.conv2d(tile_stats=ts, **kw, **common)
            res[G] = (out.float().cpu().numpy().copy(), None if ts is None else ts.cpu().numpy().copy())
        for G in (1, 2, 4, 8):
            assert np.array_equal(res[G][0], res[0][0]), (N, H, W, dil, G, "output")
            if stats:
                assert np.array_equal(res[G][1], res[0][1]), (N, H, W, dil, G, "tile records")
        assert np.abs(res[0][0]).max() > 0.1
    # ---- AdaIN consumer: records of a dilation-dA producer -> conv of dilation dB, per-sample gamma/beta ----
    for N, H, W, dA, dB in [(2, 17, 11, 1, 2), (2, 9, 20, 4, 1)]:
        C = unit
        x = g.standard_normal((N, C, H, W)).astype(np.float32)
        wA = (g.standard_normal((C, C, 3, 3)) / math.sqrt(C * 9)).astype(np.float32)
        wB = (g.standard_normal((48, C, 3, 3)) / math.sqrt(C * 9)).astype(np.float32)
        gb = (1.0 + 0.3 * g.standard_normal((N, 2 * C))).astype(np.float32)
        xt = K.nchw_to_nhwc(T(x, dev), dtype)
        wpA, wpB = K.pack_conv_weight(T(wA, dev), out_dtype=dtype), K.pack_conv_weight(T(wB, dev), out_dtype=dtype)
        sA, sB = K.conv_weight_stream(wpA), K.conv_weight_stream(wpB)
        outs = {}
        for G in (0, 2, 4):
            monkeypatch.setenv("VT_FULLKW", "0" if G == 0 else "1")
            monkeypatch.setenv("VT_FULLKW_MIN_G", "1")
            monkeypatch.setenv("VT_FULLKW_G", str(max(G, 1)))
            ts = torch.zeros(K.conv_tile_stats_bytes(N, H, W, dA, C) // 4, dtype=torch.float32, device=dev)
            yA = torch.zeros((N, H, W, C), dtype=dtype, device=dev)
            K.conv2d(src0=xt, c0=C, ld0=C, n=N, h=H, w=W, out_h=H, out_w=W, weight=wpA, weight_stream=sA, cout=C, kh=3,
                     kw=3, pad=dA, dil=dA, act=L, gain=2 ** 0.5, out=yA, ld_out=C, dtype=K.dt_code(dtype), tile_stats=ts)
            yB = torch.zeros((N, H, W, 48), dtype=dtype, device=dev)
            K.conv2d(src0=yA, c0=C, ld0=C, n=N, h=H, w=W, out_h=H, out_w=W, weight=wpB, weight_stream=sB, cout=48, kh=3,
                     kw=3, pad=dB, dil=dB, out=yB, ld_out=48, dtype=K.dt_code(dtype), tile_hint=P4, in_tile_stats=ts,
                     in_stats_dil=dA, in_gb=T(gb, dev), in_ld_gb=2 * C)
            outs[G] = yB.float().cpu().numpy().copy()
        assert np.array_equal(outs[2], outs[0]) and np.array_equal(outs[4], outs[0]), (N, H, W, dA, dB)
        assert np.abs(outs[0]).max() > 0.1


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_transpose_blur_persistent_form(dev, dtype, monkeypatch):
    """Single-chunk layers with many tiles (the 1024^2 level) run persistent workgroups: weights resident in LDS,
    next tile's patch prefetched during the blur.  Forced here on a small image (3 workgroups walk 12 tiles, a
    ragged last round) and compared bit-for-bit with the one-tile-per-workgroup form."""
    g = np.random.default_rng(5)
    cin = 32 if dtype == torch.float32 else 64
    N, H, W, cout = 2, 21, 19, 32
    x = g.standard_normal((N, cin, H, W)).astype(np.float32)
    w = (g.standard_normal((cout, cin, 3, 3)) / math.sqrt(cin * 9)).astype(np.float32)
    k1 = np.array([1, 3, 3, 1], np.float32)
    fir = T((np.outer(k1, k1) / 16.0).astype(np.float32), dev)
    xt = K.nchw_to_nhwc(T(x, dev), dtype)
    wp = K.pack_conv_weight(T(w, dev), out_dtype=dtype)
    outs = []
    for persist in ("0", "p8", "p8"):   # (the 4-wave persistent form lost to two plain workgroups per CU and was removed)
        monkeypatch.setenv("VT_UPBLUR_P8", "1" if persist == "p8" else "0")   # 16 x 16 quads, 8 waves (bf16 only)
        monkeypatch.setenv("VT_UPBLUR_WGS", "5")
        out = torch.zeros((N, 2 * H, 2 * W, cout), dtype=dtype, device=dev)
        K.conv2d(src0=xt, c0=cin, ld0=cin, n=N, h=H, w=W, out_h=2 * H, out_w=2 * W, weight=wp, cout=cout, kh=3, kw=3,
                 bias=T(g.standard_normal(cout).astype(np.float32) * 0 + 0.1, dev), act=K.ACT_LRELU, gain=2 ** 0.5,
                 out=out, ld_out=cout, dtype=K.dt_code(dtype), up_fir=fir, tile_hint=32)
        outs.append(out.float().cpu().numpy())
    assert np.abs(outs[0]).max() > 0.1 and np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


def test_conv_transpose_blur_rows_form(dev, monkeypatch):
    """conv_upblur_rows.hpp (bf16, Cin = 64 / 128, the two top levels of a frame): one wave per 28-column strip, horizontal blur
    as an MFMA on the finished accumulators, vertical blur in registers.  Forced here (VT_UPBLUR_ROWS=1) on small images: against
    the oracle's conv_transpose2d -> upfirdn2d -> fused_leaky_relu, against the tile kernel of conv_upblur.hpp (same rounding
    points: a few ulps of bf16 apart), several units per wave / one unit per wave (the same bits: the partition does not reach the
    arithmetic), a frame inside a batch = the frame alone, a cout that is not a multiple of the tile, a FIR whose taps are not
    bf16 numbers (remainder product)."""
    dtype = torch.bfloat16
    g = np.random.default_rng(77)
    k1 = np.array([1, 3, 3, 1], np.float32)
    fir_std = (np.outer(k1, k1) / 64.0 * 4.0).astype(np.float32)
    k2 = np.array([0.9, 3.1, 2.7, 1.3], np.float32)
    fir_odd = (np.outer(k1, k2) / 64.0 * 4.0).astype(np.float32)
    for N, cin, H, W, cout, fir, wgs in [(2, 64, 21, 19, 32, fir_std, "3"), (1, 128, 13, 31, 40, fir_std, "2"),
                                         (1, 64, 9, 45, 64, fir_odd, "1")]:
        x = g.standard_normal((N, cin, H, W)).astype(np.float32)
        w = (g.standard_normal((cout, cin, 3, 3)) / math.sqrt(cin * 9)).astype(np.float32)
        b = g.standard_normal(cout).astype(np.float32)
        xt = K.nchw_to_nhwc(T(x, dev), dtype)
        wp = K.pack_conv_weight(T(w, dev), out_dtype=dtype)
        xq = xt.float().cpu().permute(0, 3, 1, 2).numpy()
        wq = wp.float().cpu().numpy().reshape(cout, 3, 3, cin).transpose(0, 3, 1, 2)
        z = O.conv_transpose2d(xq, wq.transpose(1, 0, 2, 3), stride=2)
        ref = O.fused_leaky_relu(O.upfirdn2d(z, fir, pad=(1, 1)), b)

        def run(xin, n, rows, wg=None, hint=32):
            monkeypatch.setenv("VT_UPBLUR_ROWS", rows)
            if wg:
                monkeypatch.setenv("VT_UPBLUR_WGS", wg)
            else:
                monkeypatch.delenv("VT_UPBLUR_WGS", raising=False)
            out = torch.zeros((n, 2 * H, 2 * W, cout), dtype=dtype, device=dev)
            K.conv2d(src0=xin, c0=cin, ld0=cin, n=n, h=H, w=W, out_h=2 * H, out_w=2 * W, weight=wp, cout=cout, kh=3, kw=3,
                     bias=T(b, dev), act=K.ACT_LRELU, gain=2 ** 0.5, out=out, ld_out=cout, dtype=K.dt_code(dtype),
                     up_fir=T(fir, dev), tile_hint=hint)
            return out
        y_rows = run(xt, N, "1", wgs)
        import ctypes
        from vtoonify_amd import _lib
        code = _lib.lib().vt_conv2d_tile(ctypes.byref(K.make_conv_desc(
            src0=xt, c0=cin, ld0=cin, n=N, h=H, w=W, out_h=2 * H, out_w=2 * W, weight=wp, cout=cout, kh=3, kw=3, bias=T(b, dev),
            act=K.ACT_LRELU, gain=2 ** 0.5, out=y_rows, ld_out=cout, dtype=K.dt_code(dtype), up_fir=T(fir, dev), tile_hint=32)))
        assert code // 100000000 == 9 and code % 1000000 == 28032, code     # reported as its own plan kind (28-column strips)
        y = y_rows.float().cpu().permute(0, 3, 1, 2).numpy()
        assert rel_err(y, ref) < 1.2e-2, (N, cin, H, W, cout)
        y_tile = run(xt, N, "0")
        assert rel_err(y, y_tile.float().cpu().permute(0, 3, 1, 2).numpy()) < 8e-3, (N, cin, H, W, cout)
        assert torch.equal(run(xt, N, "1"), y_rows), "one unit per wave"
        # the plan's tile width follows the batch (16 channels for few-tile launches): it must not decide between this kernel
        # and the tile kernels (round 5: it did, and a frame of a video batch differed from the frame alone)
        assert torch.equal(run(xt, N, "1", wgs, hint=16), y_rows), "tile hint 16"
        if N > 1:
            assert torch.equal(run(xt[1:].contiguous(), 1, "1", wgs)[0], y_rows[1]), "frame alone"
    monkeypatch.delenv("VT_UPBLUR_ROWS")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_transpose_blur_kernel(dev, dtype, monkeypatch):
    """vt_conv_desc.up_fir: conv_transpose2d(3x3, stride 2) on the matrix cores + the 4x4 FIR blur from LDS + bias
    + LeakyReLU in one kernel (the up-sampling StyledConv at the reference's MAC count, model.py:273-286), vs the
    oracle's conv_transpose2d -> upfirdn2d(pad (1,1)) -> fused_leaky_relu on the same rounded operands, and vs the
    polyphase (phases = 4) form.  Odd sizes (partial tiles, several tiles per axis), batch 2, one / several channel
    chunks, 32- and 16-channel tiles, a cout that is not a multiple of the tile."""
    import ctypes
    from vtoonify_amd import _lib
    g = np.random.default_rng(123)
    t = F32_TOL if dtype == torch.float32 else 1.2e-2
    unit = 32 if dtype == torch.float32 else 64
    k1 = np.array([1, 3, 3, 1], np.float32)
    fir = (np.outer(k1, k1) / 64.0 * 4.0).astype(np.float32)       # make_kernel([1,3,3,1]) * factor^2 (model.py:66,192-198)
    for N, cin, H, W, cout, hint in [(2, 2 * unit, 11, 16, 40, 32), (1, unit, 13, 3, 32, 0),
                                     (1, 5 * unit, 6, 9, 32, 16),    # 16-channel tiles, >= 4 chunks: double-buffered
                                     (1, 2 * unit, 27, 15, 32, 32),   # two rows of tall tiles (44 output rows each)
                                     (2, 2 * unit, 33, 40, 40, 32)]:  # 5 x 2 flat tiles (16 x 64 output pixels each)
        x = g.standard_normal((N, cin, H, W)).astype(np.float32)
        w = (g.standard_normal((cout, cin, 3, 3)) / math.sqrt(cin * 9)).astype(np.float32)
        b = g.standard_normal(cout).astype(np.float32)
        xt = K.nchw_to_nhwc(T(x, dev), dtype)
        wp = K.pack_conv_weight(T(w, dev), out_dtype=dtype)          # plain [cout][a*3+b][cin]
        xq = xt.float().cpu().permute(0, 3, 1, 2).numpy()
        wq = wp.float().cpu().numpy().reshape(cout, 3, 3, cin).transpose(0, 3, 1, 2)
        z = O.conv_transpose2d(xq, wq.transpose(1, 0, 2, 3), stride=2)
        ref = O.fused_leaky_relu(O.upfirdn2d(z, fir, pad=(1, 1)), b)
        out = torch.zeros((N, 2 * H, 2 * W, cout), dtype=dtype, device=dev)
        kw = dict(src0=xt, c0=cin, ld0=cin, n=N, h=H, w=W, out_h=2 * H, out_w=2 * W, weight=wp, cout=cout, kh=3, kw=3,
                  bias=T(b, dev), act=K.ACT_LRELU, gain=2 ** 0.5, out=out, ld_out=cout, dtype=K.dt_code(dtype),
                  up_fir=T(fir, dev), tile_hint=hint)
        monkeypatch.setenv("VT_UPBLUR_FLAT", "0")   # (the flat tiles of the deep levels have their own block below)
        code = _lib.lib().vt_conv2d_tile(ctypes.byref(K.make_conv_desc(**kw)))
        assert code // 100000000 == 5 and code % 1000 == (hint or 16), code   # few tiles: the heuristic takes 16
        K.conv2d(**kw)
        y = out.float().cpu().permute(0, 3, 1, 2).numpy()
        assert tuple(y.shape) == ref.shape
        assert rel_err(y, ref) < t, (N, cin, H, W, cout, hint)
        # the other tile width gives the same bits (the width is chosen from the batch size: a frame inside a batch must
        # equal the frame alone)
        out_w = torch.zeros_like(out)
        K.conv2d(**{**kw, "out": out_w, "tile_hint": 32 if (hint or 16) == 16 else 16})
        assert torch.equal(out_w, out), (N, cin, H, W, cout, hint)
        if dtype == torch.bfloat16 and cin >= 2 * unit:      # ... and so do the tall tiles (24 x 16 quads, 8 waves)
            monkeypatch.setenv("VT_UPBLUR_TALL", "1")
            monkeypatch.setenv("VT_UPBLUR_DB", "99")
            out_t = torch.zeros_like(out)
            K.conv2d(**{**kw, "out": out_t, "tile_hint": 32})
            monkeypatch.delenv("VT_UPBLUR_TALL")
            monkeypatch.delenv("VT_UPBLUR_DB")
            assert torch.equal(out_t, out), (N, cin, H, W, cout, "tall")
        if dtype == torch.bfloat16:      # ... and the flattened 10 x 34-quad tiles of the deep levels (conv_upblur_flat.hpp)
            monkeypatch.setenv("VT_UPBLUR_FLAT", "1")
            code = _lib.lib().vt_conv2d_tile(ctypes.byref(K.make_conv_desc(**{**kw, "tile_hint": 32})))
            assert code // 100000000 == 10, code
            out_f = torch.zeros_like(out)
            K.conv2d(**{**kw, "out": out_f, "tile_hint": 32})
            assert torch.equal(out_f, out), (N, cin, H, W, cout, "flat")
            monkeypatch.setenv("VT_UPBLUR_FLAT_CN", "32" if code % 1000 == 16 else "16")   # ... of either width
            out_f.zero_()
            K.conv2d(**{**kw, "out": out_f, "tile_hint": 32})
            monkeypatch.delenv("VT_UPBLUR_FLAT_CN")
            monkeypatch.setenv("VT_UPBLUR_FLAT", "0")
            assert torch.equal(out_f, out), (N, cin, H, W, cout, "flat, other width")
        # the polyphase form of the same layer (blur folded into four 3x3 filters)
        wpp = K.modulate_weight(T(w, dev), torch.ones(cin, device=dev), 1.0, False, fir=T(fir, dev), out_dtype=dtype)
        out2 = torch.zeros_like(out)
        K.conv2d(src0=xt, c0=cin, ld0=cin, n=N, h=H, w=W, out_h=H, out_w=W, weight=wpp, cout=cout, kh=3, kw=3, pad=1,
                 phases=4, bias=T(b, dev), act=K.ACT_LRELU, gain=2 ** 0.5, out=out2, ld_out=cout, dtype=K.dt_code(dtype))
        assert rel_err(y, out2.float().cpu().permute(0, 3, 1, 2).numpy()) < (t if dtype == torch.float32 else 2.5e-2)
