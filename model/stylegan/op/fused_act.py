"""model.stylegan.op.fused_act (reference: model/stylegan/op/fused_act.py:87-119)."""
from vtoonify_amd.op.fused_act import FusedLeakyReLU, fused_leaky_relu  # noqa: F401
