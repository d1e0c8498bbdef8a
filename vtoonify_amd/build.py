"""Build libvtoonify_amd.so for gfx950 with hipcc (in-tree, no torch involved).

    python -m vtoonify_amd.build            # incremental
    python -m vtoonify_amd.build --force

The library is pure HIP + a C ABI (include/vtoonify_amd.h); PyTorch only ever sees raw
device pointers through ctypes (vtoonify_amd/_lib.py).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIBNAME = "libvtoonify_amd.so"
SOURCES = ["capi.hip", "fused_bias_act.hip", "upfirdn2d.hip", "style_ops.hip", "conv_igemm.hip",
           "norm_glue.hip", "frame_io.hip", "parsing_glue.hip", "raft_corr.hip", "flow_ops.hip"]
ARCH = "gfx950"
# -fno-slp-vectorize: hipcc's SLP vectoriser packs neighbouring fp32 operations into v_pk_{mul,add,fma}_f32 with lane selects;
# the form `op_sel:[0,1]` (low result reads the HIGH register of src1) returns src1.hi as ZERO on gfx950 in ~1 % of executions
# while another wave issues v_mfma_f32_16x16x32_bf16 on the same SIMD -- the wrong image rows of round 4 (DESIGN.md 4.1n,
# tools/probe/pk_war_probe.hip is the minimal reproducer).  Without the pass the library holds 70 packed fp32 instructions
# (34 000 with it), none with op_sel; tests/test_isa_lint.py pins that.  Frame rate: unchanged (profiles/r05_torgb_defect.txt).
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result", "-ffp-contract=off",
         "-fno-slp-vectorize"]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libvtoonify_amd.so)")


def lib_path() -> str:
    return os.path.join(LIBDIR, LIBNAME)


def _deps(src: str):
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "vtoonify_amd.h"))
    return [src] + hdrs


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _stamp(hipcc: str, flags) -> str:
    """What the objects were built WITH: the flags and the compiler.  A change of FLAGS alone (-fno-slp-vectorize is the fix
    for a wrong-result defect) must not leave objects of the old flags linked into the product."""
    import hashlib
    try:
        ver = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout
    except OSError:
        ver = ""
    return hashlib.sha256((" ".join(flags) + "\n" + ver).encode()).hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hipcc = hipcc_path()
    flags = list(FLAGS)
    stamp_file = os.path.join(objdir, "flags.stamp")
    stamp = _stamp(hipcc, flags)
    if not (os.path.exists(stamp_file) and open(stamp_file).read() == stamp):
        force = True    # other flags or another compiler: every object is stale
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        if force or _stale(obj, _deps(src)):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc, "-x", "hip"] + flags + ["-c", src, "-o", obj]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(compile_one, jobs))
    with open(stamp_file, "w") as f:
        f.write(stamp)
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    out = lib_path()
    if force or jobs or _stale(out, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", out] + objs
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return out


def build_variant(tag: str, defines, extra_flags=(), sources=("conv_igemm.hip",), verbose: bool = True) -> str:
    """An EXPERIMENT build of the library (tools/flake_diag.py): `sources` recompiled with -D<defines> /
    `extra_flags` (appended to the product's flags: `-fslp-vectorize` gives back round 4's packed code, the defect's reproducer)
    into build/<tag>/, linked with the product's other objects into lib/libvtoonify_amd_<tag>.so.  The product never loads
    such a file; tools select it with FLAKE_LIB / _lib.use_library()."""
    build(verbose=verbose)   # the product objects the variant links against
    hipcc = hipcc_path()
    objdir = os.path.join(HERE, "build", tag)
    os.makedirs(objdir, exist_ok=True)
    flags = list(FLAGS) + [f"-D{d}" for d in defines] + list(extra_flags)
    objs = []
    for s in SOURCES:
        if s in sources:
            obj = os.path.join(objdir, s.replace(".hip", ".o"))
            cmd = [hipcc, "-x", "hip"] + flags + ["-c", os.path.join(CSRC, s), "-o", obj]
            if verbose:
                print("[build]", " ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        else:
            obj = os.path.join(HERE, "build", s.replace(".hip", ".o"))
        objs.append(obj)
    out = os.path.join(LIBDIR, f"libvtoonify_amd_{tag}.so")
    subprocess.run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", out] + objs, check=True)
    return out


if __name__ == "__main__":
    if "--variant" in sys.argv:   # python -m vtoonify_amd.build --variant TAG [-DNAME=V ...] [-- extra hipcc flags]
        i = sys.argv.index("--variant")
        rest = sys.argv[i + 2:]
        extra = rest[rest.index("--") + 1:] if "--" in rest else []
        rest = rest[:rest.index("--")] if "--" in rest else rest
        print(build_variant(sys.argv[i + 1], [a[2:] for a in rest if a.startswith("-D")], extra))
    else:
        print(build(force="--force" in sys.argv))
