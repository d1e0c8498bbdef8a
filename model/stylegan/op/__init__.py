"""model.stylegan.op (reference: model/stylegan/op/__init__.py:1-2) -> the gfx950 operator surface."""
from .fused_act import FusedLeakyReLU, fused_leaky_relu  # noqa: F401
from .upfirdn2d import upfirdn2d  # noqa: F401  (the function shadows the submodule, as in the reference)
from . import conv2d_gradfix  # noqa: F401

__all__ = ["FusedLeakyReLU", "fused_leaky_relu", "upfirdn2d", "conv2d_gradfix"]
