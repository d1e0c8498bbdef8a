"""model.stylegan.op.conv2d_gradfix (reference: model/stylegan/op/conv2d_gradfix.py:9-75).

The module globals `enabled` / `weight_gradients_disabled` and the context manager live in
vtoonify_amd.op.conv2d_gradfix; this module forwards attribute access so that
`conv2d_gradfix.enabled = ...` assignments made by callers are seen by the implementation."""
import sys

from vtoonify_amd.op import conv2d_gradfix as _impl

sys.modules[__name__] = _impl
