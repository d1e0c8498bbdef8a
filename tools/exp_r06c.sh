#!/bin/bash
# round 6: the command-line driver on the real GPU (synthetic weights, .npy clip): one process, then under torch.distributed.run (RCCL, world size 1)
cd $GRAFT_REPO_ROOT
python - <<'PY'
import numpy as np
g = np.random.default_rng(0)
np.save("/tmp/clip.npy", g.integers(0, 256, (26, 256, 256, 3), dtype=np.uint8))
np.save("/tmp/maps.npy", (g.standard_normal((26, 19, 256, 256)) * 4).astype(np.float32))
PY
A="--content /tmp/clip.npy --video --ckpt synthetic --faceparsing_path synthetic --style_encoder_path synthetic --batch_size 4 --precision bf16"
echo "== one process, parsing maps from BiSeNet on the GPU, style code from pSp on the first frame"
python tools/style_transfer_amd.py $A --output_path /tmp/o1 2>&1 | grep -v "^[a-z_]*: \|amdgpu" | tail -6
echo "== the same with --parsing_map_path"
python tools/style_transfer_amd.py $A --parsing_map_path /tmp/maps.npy --output_path /tmp/o2 2>&1 | tail -2
echo "== torch.distributed.run, 1 rank (RCCL init + broadcast path)"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 tools/style_transfer_amd.py $A --parsing_map_path /tmp/maps.npy --output_path /tmp/o3 2>&1 | tail -2
python - <<'PY'
import numpy as np
a, b = np.load("/tmp/o2/clip_vtoonify_d.npy"), np.load("/tmp/o3/clip_vtoonify_d.npy")
print("launcher run == plain run:", a.shape, bool(np.array_equal(a, b)), "| frames differ from each other:", bool((a[0] != a[5]).any()))
PY
echo "== bench.py under the launcher (1 rank)"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29532 bench.py --gpus 1 --steps 20 --warmup 3 --no-extras --no-video --no-cpu-baseline 2>/dev/null | grep '"metric"' | python tools/bench_summary.py | head -2 | cut -c1-200
