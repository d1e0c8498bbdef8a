"""model.stylegan.op.upfirdn2d (reference: model/stylegan/op/upfirdn2d.py:149-165)."""
from vtoonify_amd.op.upfirdn2d import upfirdn2d  # noqa: F401
