#!/usr/bin/env python
"""Registers / scratch (spill) bytes / LDS of every gfx950 kernel in the built objects (vtoonify_amd/build/*.o):
the .hip_fatbin section is unbundled and the code object's metadata notes are read.

    python tools/kernel_resources.py [substring ...]      # table; `kernel_table()` is used by tests/test_abi.py

A spill in an unrolled epilogue is invisible in the source and cost one tile kernel 35 % this round.
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_table(obj_dir=None):
    """{demangled-ish kernel symbol: {"vgpr": n, "agpr": n, "scratch": bytes, "lds": bytes, "sgpr": n}}"""
    obj_dir = obj_dir or os.path.join(REPO, "vtoonify_amd", "build")
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for f in sorted(os.listdir(obj_dir)):
            if not f.endswith(".o"):
                continue
            fat, co = os.path.join(tmp, f + ".fat"), os.path.join(tmp, f + ".co")
            r = subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", os.path.join(obj_dir, f)],
                               capture_output=True)
            if r.returncode != 0 or not os.path.exists(fat):
                continue
            subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True, capture_output=True)
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
            for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
                blk = ".agpr_count:" + blk
                g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))
                name = re.search(r"\.name:\s+(\S+)", blk).group(1)
                out[name] = {"vgpr": g("vgpr_count"), "agpr": g("agpr_count"), "sgpr": g("sgpr_count"),
                             "scratch": g("private_segment_fixed_size"), "lds": g("group_segment_fixed_size")}
    return out


if __name__ == "__main__":
    t = kernel_table()
    for k, v in sorted(t.items()):
        if len(sys.argv) > 1 and not any(s in k for s in sys.argv[1:]):
            continue
        print(f"{k[:100]:<100} vgpr {v['vgpr']:4d} agpr {v['agpr']:4d} scratch {v['scratch']:5d} lds {v['lds']:7d}")
