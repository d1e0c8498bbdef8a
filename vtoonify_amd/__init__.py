"""vtoonify_amd -- MI355X-native (gfx950) implementation of VToonify's per-frame
inference hot path (model/vtoonify.py:210-277 of williamyang1991/VToonify) behind
the reference's own operator surface (model/stylegan/op).

Layout
  csrc/        hand-written HIP kernels + the C-ABI (include/vtoonify_amd.h)
  _lib.py      ctypes binding of libvtoonify_amd.so (fails loudly when missing)
  op/          drop-in for model.stylegan.op: upfirdn2d, fused_leaky_relu,
               FusedLeakyReLU, conv2d_gradfix
  engine.py    whole-frame executor (NHWC activations, MFMA implicit-GEMM convs)
  vtoonify.py  VToonify nn.Module with the reference's state_dict schema / signature
  frames.py    frame-parallel video sharding over RCCL
  synth.py     deterministic synthetic weights / frames (tests + bench)

Nothing here imports the CPU oracle (oracle/): that is test infrastructure only.
"""

__version__ = "0.1.0"
