"""Build the host-emulation twin of libvtoonify_amd.so (TEST INFRASTRUCTURE ONLY).

The same kernel sources (vtoonify_amd/csrc/*.hip) are compiled as plain C++ with
tests/emu/hip_emu.hpp force-included and -DVT_EMU, so CPU-only tests can exercise the
real index arithmetic / LDS tiling / epilogues through the real C ABI.  The product
loader (vtoonify_amd/_lib.py) never looks at this library; tests inject it explicitly.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(REPO, "vtoonify_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libvtoonify_amd_emu.so")


def clangxx():
    for c in ("/opt/rocm/lib/llvm/bin/clang++", "/opt/rocm/llvm/bin/clang++"):
        if os.path.exists(c):
            return c
    raise RuntimeError("host clang++ (ROCm LLVM) not found")


def build(force=False):
    from vtoonify_amd.build import SOURCES
    os.makedirs(OUT, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(HERE, "hip_emu.cpp")]
    deps = srcs + [os.path.join(HERE, "hip_emu.hpp"), os.path.join(REPO, "include", "vtoonify_amd.h")] + \
        [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(OUT, os.path.basename(s) + ".o")
        cmd = [clangxx(), "-x", "c++", "-std=c++17", "-O2", "-g0", "-fPIC", "-DVT_EMU", "-march=native",
               "-ffp-contract=off", "-include", os.path.join(HERE, "hip_emu.hpp"), "-I", HERE,
               "-Wno-unknown-pragmas", "-Wno-unused-value", "-c", s, "-o", o]
        procs.append((subprocess.Popen(cmd), cmd))
        objs.append(o)
    for p, cmd in procs:
        if p.wait() != 0:
            raise RuntimeError("emu build failed: " + " ".join(cmd))
    subprocess.run([clangxx(), "-shared", "-fPIC", "-o", LIB] + objs + ["-lpthread"], check=True)
    return LIB


if __name__ == "__main__":
    import sys
    sys.path.insert(0, REPO)
    print(build(force="--force" in sys.argv))
