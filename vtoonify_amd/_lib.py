"""ctypes binding of libvtoonify_amd.so -- the C ABI of include/vtoonify_amd.h.

The product loads exactly one library: vtoonify_amd/lib/libvtoonify_amd.so built for
gfx950 by `python -m vtoonify_amd.build`.  If it is missing this module raises -- there is
no CPU / eager-PyTorch fallback anywhere in the package.

(Tests may inject the host-emulation build of the same sources with `use_library(path)`;
that path is never consulted implicitly.)
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(HERE, "lib", "libvtoonify_amd.so")

ABI_VERSION = 5   # VT_ABI_VERSION of include/vtoonify_amd.h
VT_F32, VT_BF16, VT_F16 = 0, 1, 2
VT_F64 = 4     # vt_upfirdn2d / vt_fused_bias_act only: double tensors, taps and arithmetic (include/vtoonify_amd.h)
VT_F32X3 = 3   # vt_conv_desc.dtype only: fp32 tensors, products as three bf16 MFMAs (include/vtoonify_amd.h)
ACT_NONE, ACT_LRELU, ACT_RELU_TANH, ACT_SIGMOID, ACT_TANH = 0, 1, 2, 3, 4
OUT_NHWC, OUT_NCHW = 0, 1


class VtError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [
        ("src0", C.c_void_p), ("src1", C.c_void_p),
        ("c0", C.c_int32), ("c1", C.c_int32), ("ld0", C.c_int32), ("ld1", C.c_int32),
        ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
        ("out_h", C.c_int32), ("out_w", C.c_int32),
        ("weight", C.c_void_p),
        ("cout", C.c_int32),
        ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("dil", C.c_int32),
        ("phases", C.c_int32), ("transposed", C.c_int32),
        ("in_scale", C.c_void_p), ("in_shift", C.c_void_p),
        ("bias", C.c_void_p),
        ("act", C.c_int32),
        ("slope", C.c_float), ("gain", C.c_float), ("alpha", C.c_float), ("beta", C.c_float),
        ("alpha_dev", C.c_void_p),
        ("resid", C.c_void_p), ("ld_res", C.c_int32),
        ("out", C.c_void_p), ("ld_out", C.c_int32),
        ("out_layout", C.c_int32), ("out_dtype", C.c_int32), ("dtype", C.c_int32),
        ("tile_hint", C.c_int32),
        ("splitk_ws", C.c_void_p), ("splitk_ws_bytes", C.c_int64),
        ("slope_vec", C.c_void_p),
        ("rgb_weight", C.c_void_p), ("rgb_bias", C.c_void_p), ("rgb_resid", C.c_void_p), ("rgb_out", C.c_void_p),
        ("splitk_phase", C.c_int32),
        ("stats_part", C.c_void_p),
        ("post_relu", C.c_int32),
        ("weight_stream", C.c_void_p),
        ("tile_stats", C.c_void_p), ("in_tile_stats", C.c_void_p), ("in_stats_dil", C.c_int32),
        ("in_gb", C.c_void_p), ("in_ld_gb", C.c_int32),
        ("up_fir", C.c_void_p), ("pad_w_p1", C.c_int32), ("rgb_only", C.c_int32), ("in_absdiff", C.c_int32),
    ]


class LinearItem(C.Structure):
    _fields_ = [("y", C.c_void_p), ("x", C.c_void_p), ("W", C.c_void_p), ("b", C.c_void_p),
                ("ld_y", C.c_int32), ("ld_x", C.c_int32), ("rows", C.c_int32), ("in_dim", C.c_int32),
                ("out_dim", C.c_int32), ("act", C.c_int32),
                ("w_scale", C.c_float), ("b_scale", C.c_float), ("slope", C.c_float), ("gain", C.c_float)]


class ModulateItem(C.Structure):
    _fields_ = [("out", C.c_void_p), ("weight", C.c_void_p), ("s", C.c_void_p), ("fir", C.c_void_p),
                ("cout", C.c_int32), ("cin", C.c_int32), ("k", C.c_int32), ("demodulate", C.c_int32),
                ("scale", C.c_float), ("reserved", C.c_int32)]


_SIGS = {
    "vt_abi_version": (C.c_int, []),
    "vt_last_error": (C.c_char_p, []),
    "vt_build_target": (C.c_char_p, []),
    "vt_upfirdn2d_out_size": (C.c_int, [C.c_int] * 12 + [C.POINTER(C.c_int)] * 2),
    "vt_upfirdn2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64] + [C.c_int] * 12 +
                     [C.c_int, C.c_void_p]),
    "vt_fused_bias_act": (C.c_int, [C.c_void_p] * 4 + [C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int,
                                                       C.c_float, C.c_float, C.c_int, C.c_void_p]),
    "vt_conv2d": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    "vt_conv2d_tile": (C.c_int, [C.POINTER(ConvDesc)]),
    "vt_conv2d_ws_bytes": (C.c_int64, [C.POINTER(ConvDesc)]),
    "vt_conv2d_splitk_mode": (C.c_int, [C.POINTER(ConvDesc)]),
    "vt_conv_weight_stream_bytes": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "vt_conv_tile_stats_bytes": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "vt_conv_weight_stream": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vt_pack_conv_weight": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_void_p]),
    "vt_modulate_weight": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                     C.c_float, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "vt_linear": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                            C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float,
                            C.c_void_p]),
    "vt_linear_batch": (C.c_int, [C.POINTER(LinearItem), C.c_int, C.c_void_p]),
    "vt_modulate_weight_batch": (C.c_int, [C.POINTER(ModulateItem), C.c_int, C.c_int, C.c_void_p]),
    "vt_pixel_norm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "vt_style_gate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "vt_linear_batch_gated": (C.c_int, [C.POINTER(LinearItem), C.c_int, C.c_void_p, C.c_void_p]),
    "vt_modulate_weight_batch_gated": (C.c_int, [C.POINTER(ModulateItem), C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vt_pixel_norm_gated": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vt_instnorm_ws_bytes": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "vt_instnorm_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                    C.c_void_p]),
    "vt_instnorm_apply": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                    C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "vt_frame_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int,
                                C.c_int, C.c_void_p]),
    "vt_frame_unpack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vt_instnorm_plane": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "vt_instnorm_apply_stats": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                    C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "vt_affine_apply": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                  C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vt_channel_mean": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                  C.c_void_p]),
    "vt_se_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 8 + [C.c_void_p]),
    "vt_upsample_bilinear_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 7 + [C.c_void_p]),
    "vt_maxpool2d": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 8 + [C.c_void_p]),
    "vt_gate_add_nearest": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 7 + [C.c_void_p]),
    "vt_resize_bilinear": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p] + [C.c_int] * 10 +
                           [C.c_float, C.c_void_p]),
    "vt_corr_lookup": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 7 + [C.c_float, C.c_float, C.c_void_p]),
    "vt_avgpool2x2": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]),
    "vt_flow_warp": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]),
    "vt_parsing_fuse": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 5 + [C.c_float, C.c_void_p]),
    "vt_eltwise2": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int,
                              C.c_int, C.c_void_p]),
    "vt_gru_blend": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "vt_coords_from_flow": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vt_convex_upsample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vt_fusion_pack": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vt_nchw_to_nhwc": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.c_void_p]),
    "vt_nhwc_to_nchw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.c_void_p]),
    "vt_mfma_selftest": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGS)

_lib = None
_lib_path = None


def _bind(path: str):
    # PyTorch-ROCm ships its own libamdhip64; load it FIRST so that this library's HIP calls bind
    # to the runtime that owns torch's device context and streams.  (Loading ours first pulls
    # /opt/rocm's copy into the process and the two runtimes do not see each other's devices:
    # "no ROCm-capable device is detected" at the first launch.)
    import torch  # noqa: F401
    lib = C.CDLL(path)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.vt_abi_version() != ABI_VERSION:
        raise VtError(f"{path}: ABI version {lib.vt_abi_version()} != {ABI_VERSION}")
    return lib


def use_library(path: str):
    """Bind an explicit library file (tests: the host-emulation build)."""
    global _lib, _lib_path
    _lib = _bind(path)
    _lib_path = path
    return _lib


def lib():
    global _lib, _lib_path
    if _lib is None:
        if not os.path.exists(DEFAULT_LIB):
            raise VtError(
                f"{DEFAULT_LIB} is missing: the gfx950 HIP library has not been built. Run "
                "`python -m vtoonify_amd.build` (needs hipcc). vtoonify_amd has no CPU fallback.")
        _lib = _bind(DEFAULT_LIB)
        _lib_path = DEFAULT_LIB
    return _lib


def lib_path():
    lib()
    return _lib_path


def is_emulation() -> bool:
    return lib().vt_build_target() != b"gfx950"


def emulation_injected() -> bool:
    """True only while a TEST has bound the host-emulation build with use_library().  Never loads anything: the operator
    surface asks this to tell a CPU tensor that belongs to the emulation (tests) from a CPU tensor of a user, which takes
    the native torch formula like the reference's own CPU branch (op/upfirdn2d.py:159-160, op/fused_act.py:105-116)."""
    return _lib is not None and _lib.vt_build_target() != b"gfx950"


def release_library():
    """Forget the bound library (tests: leave the emulation); the next lib() binds the product library again."""
    global _lib, _lib_path
    _lib = None
    _lib_path = None


def check(rc: int, what: str = ""):
    if rc != 0:
        raise VtError(f"{what} failed (code {rc}): {lib().vt_last_error().decode()}")
