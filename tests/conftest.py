import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name))
    d = {k: z[k] for k in z.files}
    meta = None
    if "meta" in d:
        meta = json.loads(bytes(d.pop("meta")).decode())
    return d, meta


def load_keys(tag):
    with open(os.path.join(GOLDEN, f"keys_{tag}.json")) as f:
        return {k: tuple(v) for k, v in json.load(f).items()}


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


# ---- execution backends -----------------------------------------------------------------
# "emu": the kernel sources compiled for the host (tests/emu) -- runs in the GPU-less
#        authoring container and checks index arithmetic / tiling / epilogues / host logic.
# "gpu": the product library on a real MI355X (pytest -m gpu, via gpurun).
import torch  # noqa: E402


@pytest.fixture(params=[pytest.param("emu"), pytest.param("gpu", marks=pytest.mark.gpu)])
def dev(request):
    from vtoonify_amd import _lib
    if request.param == "emu":
        from emu import build_emu
        _lib.use_library(build_emu.build())
        return torch.device("cpu")
    assert torch.cuda.is_available(), "gpu-marked test needs a GPU"
    _lib.use_library(_lib.DEFAULT_LIB)  # raises if the gfx950 .so is missing: no silent fallback
    assert not _lib.is_emulation()
    return torch.device("cuda:0")


def psnr(a, b, data_range):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    mse = float(((a - b) ** 2).mean())
    return float("inf") if mse == 0 else 10.0 * np.log10(data_range ** 2 / mse)
