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
