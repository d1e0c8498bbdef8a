"""Frame-parallel sharding (vtoonify_amd/frames.py): pure partition logic + the N>1 path on
CPU with gloo, world_size 2 (the RCCL path is identical code with backend="nccl")."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from vtoonify_amd import frames

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n,ws", [(0, 1), (1, 8), (7, 8), (8, 8), (9, 8), (960, 8), (301, 4), (5, 2)])
def test_shard_range_partitions_in_order(n, ws):
    spans = [frames.shard_range(n, r, ws) for r in range(ws)]
    assert spans[0][0] == 0 and spans[-1][1] == n
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 == b0 and a0 <= a1
    sizes = [b - a for a, b in spans]
    assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_shard_range_rejects_bad_rank():
    with pytest.raises(ValueError):
        frames.shard_range(10, 2, 2)
    with pytest.raises(ValueError):
        frames.shard_range(-1, 0, 2)


def test_batches_ragged_tail():
    assert frames.batches(3, 14, 4) == [(3, 7), (7, 11), (11, 14)]
    assert frames.batches(5, 5, 4) == []
    with pytest.raises(ValueError):
        frames.batches(0, 4, 0)


def test_single_process_broadcast_is_identity():
    shapes = {"b": (3,), "a": (2, 5)}
    sd = {k: torch.randn(s) for k, s in shapes.items()}
    out = frames.broadcast_state_dict(shapes, sd, torch.device("cpu"), bucket_elems=4)
    for k in shapes:
        assert torch.equal(out[k], sd[k])
    s, d = frames.broadcast_style(torch.ones(1, 18, 512), 0.25, torch.device("cpu"))
    assert s.shape == (1, 18, 512) and d == 0.25


_WORKER = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["VT_REPO"])
from vtoonify_amd import frames

rank, local_rank, ws = frames.init("gloo")
assert ws == 2 and dist.get_backend() == "gloo"
dev = torch.device("cpu")
shapes = {"w.%d" % i: (i + 1, 7) for i in range(9)}
shapes["big"] = (1000,)
gen = torch.Generator().manual_seed(0)
full = {k: torch.randn(s, generator=gen) for k, s in sorted(shapes.items())}
sd = full if rank == 0 else None
got = frames.broadcast_state_dict(shapes, sd, dev, bucket_elems=300)   # forces several buckets
for k in shapes:
    assert torch.equal(got[k], full[k]), k
style = torch.arange(18 * 512, dtype=torch.float32).view(1, 18, 512)
s, d = frames.broadcast_style(style if rank == 0 else None, 0.75 if rank == 0 else None, dev)
assert torch.equal(s, style) and d == 0.75

# a 7-frame "video": each rank fills its shard with the global frame index, rank 0 gathers in order
n = 7
a, b = frames.shard_range(n, rank, ws)
local = torch.stack([torch.full((3, 4, 4), float(i)) for i in range(a, b)])
allf = frames.gather_frames(local, n, dst=0)
if rank == 0:
    assert allf.shape == (n, 3, 4, 4)
    assert torch.equal(allf[:, 0, 0, 0], torch.arange(n, dtype=torch.float32))
else:
    assert allf is None
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_gloo_world_size_2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), VT_REPO=REPO, OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o}"
        assert f"rank {r} ok" in o
