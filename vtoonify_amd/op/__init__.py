"""Drop-in for the reference's `model.stylegan.op` package
(model/stylegan/op/__init__.py:1-2): same names, same signatures, gfx950 kernels.

    from vtoonify_amd.op import FusedLeakyReLU, fused_leaky_relu, upfirdn2d, conv2d_gradfix
"""
from .fused_act import FusedLeakyReLU, fused_leaky_relu
from .upfirdn2d import upfirdn2d
from . import conv2d_gradfix

__all__ = ["FusedLeakyReLU", "fused_leaky_relu", "upfirdn2d", "conv2d_gradfix"]
