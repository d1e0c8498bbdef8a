"""Mirror of the reference's `model` package for the per-frame inference path.

`style_transfer.py:11-14` and `util.py:14-15` of williamyang1991/VToonify import

    model.vtoonify.VToonify                                   -> vtoonify_amd.vtoonify
    model.bisenet.model.BiSeNet                               -> vtoonify_amd.bisenet
    model.stylegan.op (FusedLeakyReLU, fused_leaky_relu, upfirdn2d, conv2d_gradfix)
                                                              -> vtoonify_amd.op
    model.encoder.encoders.psp_encoders.GradualStyleEncoder   -> vtoonify_amd.psp
    model.raft.core.raft.RAFT  (smooth_parsing_map.py:12)     -> vtoonify_amd.raft

Putting this repository BEFORE the reference checkout on sys.path makes those imports resolve to the
gfx950 implementations; every other submodule (`model.encoder.align_all_parallel`, `model.raft.core.utils`, the
training-only `model.stylegan.model` ...) still resolves to the reference checkout further down the
path, because each mirrored package extends its `__path__` over all same-named directories
(pkgutil.extend_path; the reference's own `__init__.py` files are empty).  Nothing here computes.
"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
