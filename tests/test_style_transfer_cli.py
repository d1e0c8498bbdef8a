"""tools/style_transfer_amd.py: the reference's command line (style_transfer.py:17-46) over the frame-parallel video driver.

* the option table equals the reference's 16 `add_argument`s (names, types, defaults) -- against the mounted reference when
  it is there, against the table below otherwise;
* one process, host emulation: the .npy video that comes out equals the frames computed one by one in the reference loop's
  order of operations (pack -> forward -> clamp -> uint8), in frame order;
* two processes (gloo, world_size 2, host emulation): rank 0 reads the weights and the style, one broadcast, every rank runs
  its contiguous shard, and the single output file is byte-identical to the one-process run
  (style_transfer.py:176-181: ordered write);
* `--cpu`: the module's eager graph over the CPU branch of the operator surface produces the same uint8 frame (+-1 level)
  as the executor in fp32.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REPO

sys.path.insert(0, os.path.join(REPO, "tools"))
import style_transfer_amd as cli  # noqa: E402

REFERENCE_OPTIONS = {   # style_transfer.py:21-37: dest -> (type or action, default)
    "content": (str, "./data/077436.jpg"), "style_id": (int, 26), "style_degree": (float, 0.5),
    "color_transfer": ("store_true", False), "ckpt": (str, "./checkpoint/vtoonify_d_cartoon/vtoonify_s_d.pt"),
    "output_path": (str, "./output/"), "scale_image": ("store_true", False),
    "style_encoder_path": (str, "./checkpoint/encoder.pt"), "exstyle_path": (str, None),
    "faceparsing_path": (str, "./checkpoint/faceparsing.pth"), "video": ("store_true", False), "cpu": ("store_true", False),
    "backbone": (str, "dualstylegan"), "padding": (int, [200, 200, 200, 200]), "batch_size": (int, 4),
    "parsing_map_path": (str, None),
}


def _table(parser):
    out = {}
    for a in parser._actions:
        if a.dest == "help":
            continue
        kind = "store_true" if type(a).__name__ == "_StoreTrueAction" else a.type
        out[a.dest] = (kind, a.default)
    return out


def test_option_table_is_the_references():
    mine = _table(cli.build_parser())
    for k, v in REFERENCE_OPTIONS.items():
        assert mine[k] == v, k
    opt = cli.parse(["--ckpt", "/x/y/model.pt"])
    assert opt.exstyle_path == "/x/y/exstyle_code.npy"          # style_transfer.py:41-42
    ref = "/root/reference/style_transfer.py"
    if os.path.exists(ref):                                      # the table above is the reference's, not a memory of it
        import ast
        tree = ast.parse(open(ref).read())
        found = {}
        for node in ast.walk(tree):
            if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "add_argument":
                kw = {k.arg: k.value for k in node.keywords}
                dest = node.args[0].value.lstrip("-")
                if "action" in kw:
                    found[dest] = ("store_true", False)
                else:
                    found[dest] = (eval(kw["type"].id), ast.literal_eval(kw["default"]))
        assert found == REFERENCE_OPTIONS


def _clip(tmp_path, n=5, H=16, W=24, seed=3):
    g = np.random.default_rng(seed)
    frames = g.integers(0, 256, (n, H, W, 3), dtype=np.uint8)
    maps = (g.standard_normal((n, 19, H, W)) * 4).astype(np.float32)
    np.save(tmp_path / "clip.npy", frames)
    np.save(tmp_path / "maps.npy", maps)
    np.save(tmp_path / "code.npy", g.standard_normal((1, 18, 512)).astype(np.float32))
    return frames, maps


def _args(tmp_path, out, backbone="toonify", extra=()):
    return ["--content", str(tmp_path / "clip.npy"), "--video", "--parsing_map_path", str(tmp_path / "maps.npy"),
            "--intrinsic_code", str(tmp_path / "code.npy"), "--ckpt", "synthetic", "--backbone", backbone,
            "--output_path", str(out), "--batch_size", "2", "--depth", "2", "--precision", "bf16", *extra]


def _emu():
    from emu import build_emu
    from vtoonify_amd import _lib
    _lib.use_library(build_emu.build())


def test_one_process_video_equals_frame_by_frame(tmp_path):
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import frames_oracle as FO
    from vtoonify_amd import synth
    _emu()
    frames, maps = _clip(tmp_path, n=3)
    rep = cli.main(_args(tmp_path, tmp_path / "out1"), device="cpu")
    assert rep["frames"] == 3 and rep["shard"] == (0, 3)
    got = np.load(rep["output"])
    assert got.shape == (3, 64, 96, 3) and got.dtype == np.uint8
    # the reference loop, one frame at a time, with the same weights and style code
    from vtoonify_amd.vtoonify import VToonify
    m = VToonify(backbone="toonify", compute_dtype=torch.bfloat16)
    m.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, 0))
    s_w = m.zplus2wplus(torch.from_numpy(np.load(tmp_path / "code.npy")))
    for i in range(3):
        x = torch.from_numpy(FO.pack_inputs(frames[i][None], maps[i][None]))
        y = m(x, s_w, d_s=None)
        assert np.array_equal(got[i], FO.tensor2cv2(y[0].float().numpy())), i


_WORKER = """
import os, sys
sys.path.insert(0, os.environ["VT_REPO"]); sys.path.insert(0, os.path.join(os.environ["VT_REPO"], "tests"))
sys.path.insert(0, os.path.join(os.environ["VT_REPO"], "tools"))
from emu import build_emu
from vtoonify_amd import _lib
_lib.use_library(build_emu.build())
import style_transfer_amd as cli
rep = cli.main(sys.argv[1:], device="cpu", backend="gloo")
print("rank", rep["rank"], "shard", rep["shard"], "ok")
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_write_the_same_video(tmp_path):
    _emu()
    _clip(tmp_path, H=16, W=16)
    one = np.load(cli.main(_args(tmp_path, tmp_path / "out1", backbone="dualstylegan"), device="cpu")["output"])
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), VT_REPO=REPO, OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)] + _args(tmp_path, tmp_path / "out2", backbone="dualstylegan"),
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o}"
    assert "shard (0, 3)" in outs[0] and "shard (3, 5)" in outs[1]
    two = np.load(tmp_path / "out2" / "clip_vtoonify_d.npy")
    assert two.shape == one.shape and np.array_equal(two, one), "sharded run must write the one-process video"


def test_cpu_flag_runs_the_eager_graph(tmp_path):
    """--cpu (style_transfer.py:32,55): no library, torch on the host; same picture as the executor in fp32 (+-1 uint8 level)."""
    from vtoonify_amd import _lib
    _clip(tmp_path, n=2, H=16, W=16)
    _emu()
    want = np.load(cli.main(_args(tmp_path, tmp_path / "o_eng", extra=("--precision", "fp32_exact")), device="cpu")["output"])
    _lib.release_library()
    got = np.load(cli.main(_args(tmp_path, tmp_path / "o_cpu", extra=("--cpu",)))["output"])
    assert _lib._lib is None, "--cpu must not load a library"
    assert got.shape == want.shape == (2, 64, 64, 3)
    assert np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 1


def test_missing_reader_is_an_error_not_a_guess(tmp_path):
    try:
        import cv2  # noqa: F401
        pytest.skip("cv2 present")
    except ImportError:
        pass
    (tmp_path / "a.mp4").write_bytes(b"")
    with pytest.raises(SystemExit):
        cli.main(["--content", str(tmp_path / "a.mp4"), "--video", "--ckpt", "synthetic"], device="cpu")
