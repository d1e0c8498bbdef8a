#!/usr/bin/env python3
"""Golden fixtures for the BiSeNet face-parsing network from the REAL reference (authoring
container only).

    python tests/golden/make_golden_bisenet.py    # writes tests/golden/bisenet.npz, keys_bisenet.json

model.bisenet.model.BiSeNet(19) is imported from /root/reference.  Two things it needs do not exist
here and are stubbed WITHOUT touching its arithmetic: `import torchvision` (imported, never used,
model/bisenet/model.py:8) and the ImageNet checkpoint download in Resnet18.init_weight
(model/bisenet/resnet.py:82-88; `modelzoo.load_url` returns an empty dict, the weights are then
replaced by the deterministic synthetic ones of vtoonify_amd.synth, which both sides regenerate
from the key names).  Also recorded: the reference's parsing pre/post-processing around the net
(style_transfer.py:171-172).  Nothing is copied from the reference: only tensors it computes.
"""
import json
import os
import sys
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("VTOONIFY_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import torch.utils.model_zoo as modelzoo  # noqa: E402

sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
modelzoo.load_url = lambda *a, **k: {}
from model.bisenet.model import BiSeNet  # noqa: E402

sys.path.append(REPO)
from vtoonify_amd import synth  # noqa: E402

torch.set_grad_enabled(False)


def main():
    net = BiSeNet(n_classes=19).eval()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    with open(os.path.join(HERE, "keys_bisenet.json"), "w") as f:
        json.dump({k: list(v) for k, v in shapes.items()}, f, indent=0)
    net.load_state_dict(synth.synth_state_dict(shapes, 0))
    g = torch.Generator().manual_seed(91)
    out = {}
    for name, (b, h, w) in {"s64": (1, 64, 64), "s96x64": (2, 96, 64)}.items():
        x = torch.rand(b, 3, h, w, generator=g) * 4 - 2   # the net sees 2 * frame, frame in [-1, 1]
        feat_res8, feat_cp8, feat_cp16 = net.cp(x)
        y, y16, y32 = net(x)
        out[name + "__x"] = x.numpy()
        out[name + "__res8"] = feat_res8.numpy()
        out[name + "__cp8"] = feat_cp8.numpy()
        out[name + "__cp16"] = feat_cp16.numpy()
        out[name + "__y"] = y.numpy()
        if name == "s64":
            out[name + "__y16"] = y16.numpy()
            out[name + "__y32"] = y32.numpy()
        print(name, tuple(y.shape), float(y.abs().max()), float(feat_cp8.abs().max()))
    # style_transfer.py:171-172: parsing maps of a frame batch x in [-1, 1]
    for name, (b, h, w) in {"frame32": (1, 32, 32), "frame40x24": (1, 40, 24)}.items():
        x = torch.rand(b, 3, h, w, generator=g) * 2 - 1
        x_p = F.interpolate(net(2 * (F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)))[0],
                            scale_factor=0.5, recompute_scale_factor=False).detach()
        out[name + "__x"] = x.numpy()
        out[name + "__xp"] = x_p.numpy()
        print(name, tuple(x_p.shape), float(x_p.abs().max()))
    path = os.path.join(HERE, "bisenet.npz")
    np.savez_compressed(path, **out)
    print(f"wrote bisenet.npz: {os.path.getsize(path) / 1024:.1f} KiB; {len(shapes)} state_dict keys")


if __name__ == "__main__":
    main()
