"""Drop-in import surface: with this repository ahead of the reference checkout on
sys.path, the reference's own `style_transfer.py` lines -- its imports (11-14), the model construction /
checkpoint loading (62-68) and the per-batch call (176) -- run UNMODIFIED on the gfx950 implementation
(host emulation here), while `model.encoder.align_all_parallel` & co still come from the reference tree.
cv2 / dlib / torchvision / wget are absent from the image and are stubbed (none of them is on the hot path).

Runs in a subprocess: the mirror package is called `model`, like the reference's, and must not leak into the
other tests' sys.modules.  Needs /root/reference (authoring container only)."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import REPO

REF = "/root/reference"

SCRIPT = r'''
import argparse, json, os, sys, types
REPO, REF, TMP, BACKBONE = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
sys.path[:0] = [REPO, os.path.join(REPO, "tests"), REF]
for name in ("cv2", "dlib", "wget", "torchvision", "torchvision.transforms"):
    sys.modules[name] = types.ModuleType(name)
tv = sys.modules["torchvision.transforms"]
tv.Compose = lambda ts: (lambda x: x)
tv.ToTensor = lambda: None
tv.Normalize = lambda mean, std: None
sys.modules["torchvision"].transforms = tv
import numpy as np
import torch
from emu import build_emu
from vtoonify_amd import _lib, synth
_lib.use_library(build_emu.build())

def lines(path, a, b):   # 1-based inclusive range of a reference source file, dedented, verbatim
    import textwrap
    with open(path) as f:
        src = f.readlines()[a - 1:b]
    return textwrap.dedent("".join(src))

ST = os.path.join(REF, "style_transfer.py")
ns = {}
exec(compile(lines(ST, 11, 14), ST, "exec"), ns)          # the four model/util imports
import model, vtoonify_amd.vtoonify, vtoonify_amd.bisenet, vtoonify_amd.psp, vtoonify_amd.op
assert ns["VToonify"] is vtoonify_amd.vtoonify.VToonify
assert ns["BiSeNet"] is vtoonify_amd.bisenet.BiSeNet
assert ns["align_face"].__module__ == "model.encoder.align_all_parallel"
assert sys.modules["model.encoder.align_all_parallel"].__file__.startswith(REF)      # not mirrored: reference code
import util
assert util.conv2d_gradfix is vtoonify_amd.op.conv2d_gradfix                        # util.py:14
assert util.GradualStyleEncoder is vtoonify_amd.psp.GradualStyleEncoder            # util.py:15
from model.stylegan.op import FusedLeakyReLU, fused_leaky_relu, upfirdn2d, conv2d_gradfix
assert upfirdn2d is vtoonify_amd.op.upfirdn2d and callable(upfirdn2d)

# synthetic checkpoints with the reference's schemas (no real checkpoints in the image)
def shapes(tag):
    with open(os.path.join(REPO, "tests", "golden", f"keys_{tag}.json")) as f:
        return {k: tuple(v) for k, v in json.load(f).items()}
sd = synth.synth_state_dict(shapes("D" if BACKBONE == "dualstylegan" else "T"), 0)
torch.save({"g_ema": sd}, os.path.join(TMP, "vtoonify.pt"))
torch.save(synth.synth_state_dict(shapes("bisenet"), 1), os.path.join(TMP, "faceparsing.pth"))
args = argparse.Namespace(backbone=BACKBONE, ckpt=os.path.join(TMP, "vtoonify.pt"),
                          faceparsing_path=os.path.join(TMP, "faceparsing.pth"), style_degree=0.5)
ns.update(args=args, device="cpu", torch=torch)
exec(compile(lines(ST, 62, 68), ST, "exec"), ns)          # VToonify(...).load_state_dict(...).to(device); BiSeNet
vt = ns["vtoonify"]
assert isinstance(vt, vtoonify_amd.vtoonify.VToonify) and isinstance(ns["parsingpredictor"], vtoonify_amd.bisenet.BiSeNet)
ns["inputs"] = synth.synth_frames(2, 16, 16, seed=3)
ns["s_w"] = synth.synth_style(seed=5)
exec(compile(lines(ST, 176, 176), ST, "exec"), ns)       # y_tilde = vtoonify(inputs, s_w.repeat(B,1,1), d_s=...)
y = ns["y_tilde"]
assert tuple(y.shape) == (2, 3, 64, 64) and bool(torch.isfinite(y).all())
# the drop-in computes in the REFERENCE'S precision unless told otherwise (fp32; bf16 is opt-in: VTOONIFY_AMD_DTYPE /
# compute_dtype): the call above, made by the reference's own line, matches the CPU oracle to the fp32 bar
assert vt.compute_dtype == torch.float32
if BACKBONE == "toonify":   # (the D forward against the oracle is tests/test_engine.py::test_golden_fp32; once is enough here)
    from oracle import vtoonify_oracle as O
    O.set_backend("torch")
    want = O.vtoonify_forward(synth.to_numpy_sd(sd), ns["inputs"].numpy(), ns["s_w"].repeat(2, 1, 1).numpy(), 0.5, BACKBONE)
    err = float((y.numpy() - want).__abs__().max() / abs(want).max())
    assert err <= 1e-4, err
# the same frames through the engine API directly: the module call is the same computation
if BACKBONE == "toonify":   # (an emulated D forward is ~10 s: once is enough)
    y2 = vt.engine().forward(ns["inputs"], ns["s_w"], 0.5)
    assert torch.equal(y, y2)
# a weight load into a SUBMODULE invalidates the packed engine (ADVICE r1)
e0 = vt.engine()
vt.generator.load_state_dict(vt.generator.state_dict())
assert vt.engine() is not e0
print("IMPORT_SURFACE_OK")
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not mounted")
@pytest.mark.parametrize("backbone", ["dualstylegan", "toonify"])
def test_reference_style_transfer_lines_run_on_the_mirror(tmp_path, backbone):
    script = tmp_path / "surface.py"
    script.write_text(SCRIPT)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, str(script), REPO, REF, str(tmp_path), backbone], capture_output=True, text=True,
                       env=env, timeout=900)
    assert r.returncode == 0 and "IMPORT_SURFACE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_mirror_package_re_exports_without_the_reference():
    """The mirror alone (no reference on the path): the six import targets resolve to vtoonify_amd."""
    code = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {REPO!r})
        from model.vtoonify import VToonify
        from model.bisenet.model import BiSeNet
        from model.stylegan.op import FusedLeakyReLU, fused_leaky_relu, upfirdn2d, conv2d_gradfix
        from model.stylegan.op.conv2d_gradfix import conv2d, conv_transpose2d, no_weight_gradients
        from model.encoder.encoders.psp_encoders import GradualStyleEncoder
        from model.raft.core.raft import RAFT                     # smooth_parsing_map.py:12
        import vtoonify_amd.op, vtoonify_amd.vtoonify, vtoonify_amd.psp, vtoonify_amd.bisenet, vtoonify_amd.raft
        assert RAFT is vtoonify_amd.raft.RAFT
        assert VToonify is vtoonify_amd.vtoonify.VToonify and BiSeNet is vtoonify_amd.bisenet.BiSeNet
        assert GradualStyleEncoder is vtoonify_amd.psp.GradualStyleEncoder
        assert conv2d_gradfix is vtoonify_amd.op.conv2d_gradfix and conv2d is conv2d_gradfix.conv2d
        assert upfirdn2d is vtoonify_amd.op.upfirdn2d
        conv2d_gradfix.enabled = False      # module globals are the implementation's own
        assert vtoonify_amd.op.conv2d_gradfix.enabled is False
        print("MIRROR_OK")
    """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert r.returncode == 0 and "MIRROR_OK" in r.stdout, r.stdout + r.stderr
