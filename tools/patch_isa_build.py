#!/usr/bin/env python
"""EXPERIMENT build with the device ISA patched by hand: csrc/conv_igemm.hip is compiled to gfx950 assembly
with -fslp-vectorize (round 4's packed code: the defect's reproducer), a rule edits the assembly, and the result is assembled,
linked, bundled and embedded into a host object exactly as hipcc does (`hipcc -###`), then linked with the product's other
objects into vtoonify_amd/lib/libvtoonify_amd_<TAG>.so.

    python tools/patch_isa_build.py TAG MODE [--asm dev.s]
      MODE war_after   s_nop 1 AFTER every v_pk_{add,mul,fma}_f32 one of whose source registers is written by the next VALU
                       instruction of the wave (fall-through and branch target)           -- removes the suspected hazard
           war_before  the same s_nop 1 in FRONT of those instructions                    -- control: same count, same kernels
           none        no edit                                                            -- control: the tool chain itself
"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
CSRC = os.path.join(REPO, "vtoonify_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result", "-ffp-contract=off",
         "-fslp-vectorize"]   # (the experiments of profiles/r05_torgb_defect.txt ran at commit af79b78, where this was -DVT_EXP=1)
_REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(text):
    s = set()
    for m in _REG.finditer(text):
        if m.group(3) is not None:
            s.add(int(m.group(3)))
        else:
            s.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return s


def is_inst(l):
    t = l.strip()
    return bool(t) and not t.startswith((";", ".", "#")) and not t.endswith(":") and l.startswith("\t")


def war_sites(lines):
    """indices of packed-fp32 instructions whose source registers the wave's next VALU instruction overwrites"""
    labels = {l.strip()[:-1]: i for i, l in enumerate(lines) if l.strip().endswith(":") and not l.startswith("\t")}

    def next_valu(i, depth=0):
        """first VALU instruction(s) at or after line i along fall-through and branch targets"""
        out = []
        while i < len(lines):
            l = lines[i]
            if is_inst(l):
                t = l.strip()
                op = t.split()[0]
                if op.startswith("v_"):
                    out.append(t)
                    return out
                if op.startswith("s_cbranch") or op == "s_branch":
                    tgt = t.split()[1]
                    if tgt in labels and depth < 2:
                        out += next_valu(labels[tgt], depth + 1)
                    if op == "s_branch":
                        return out
                if op in ("s_endpgm", "s_setpc_b64"):
                    return out
            i += 1
        return out

    sites = []
    for i, l in enumerate(lines):
        if not is_inst(l):
            continue
        t = l.strip()
        if not re.match(r"v_pk_(add|mul|fma)_f32\b", t):
            continue
        ops = t.split(None, 1)[1].split(",")
        src = regs(",".join(ops[1:]))
        pdst = regs(ops[0])
        for nv in next_valu(i + 1):
            nops = nv.split(None, 1)[1].split(",") if len(nv.split(None, 1)) > 1 else [""]
            dst = regs(nops[0])
            # a PURE write-after-read: the next VALU instruction overwrites a source of the packed instruction and does not
            # read its result (a reader would wait for the result anyway)
            if (dst & src) and not (regs(",".join(nops[1:])) & pdst):
                sites.append(i)
                break
    return sites


def main():
    tag, mode = sys.argv[1], sys.argv[2]
    work = os.path.join(REPO, "vtoonify_amd", "build", tag)
    os.makedirs(work, exist_ok=True)
    src = os.path.join(CSRC, "conv_igemm.hip")
    asm = sys.argv[sys.argv.index("--asm") + 1] if "--asm" in sys.argv else os.path.join(work, "dev.s")
    if not os.path.exists(asm):
        subprocess.run(["hipcc", "-x", "hip"] + FLAGS + ["--cuda-device-only", "-S", src, "-o", asm], check=True)
    lines = open(asm).read().split("\n")
    sites = war_sites(lines) if mode != "none" else []
    per_kernel = {}
    cur = None
    site_set = set(sites)
    out = []
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = m.group(1)
        if i in site_set:
            per_kernel[cur] = per_kernel.get(cur, 0) + 1
            if mode == "war_before":
                out.append("\ts_nop 1")
            out.append(l)
            if mode == "war_after":
                out.append("\ts_nop 1")
        else:
            out.append(l)
    patched = os.path.join(work, "dev_patched.s")
    open(patched, "w").write("\n".join(out))
    print(f"{mode}: {len(sites)} sites in {len(per_kernel)} kernels")
    for k, n in sorted(per_kernel.items(), key=lambda kv: -kv[1])[:12]:
        print(f"   {n:4d}  {k[:110]}")
    dev_o, dev_out, fb = os.path.join(work, "dev.o"), os.path.join(work, "dev.out"), os.path.join(work, "dev.hipfb")
    subprocess.run([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", patched, "-o", dev_o],
                   check=True)
    subprocess.run([f"{LLVM}/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", dev_out, dev_o],
                   check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "-type=o", "-bundle-align=4096",
                    "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null",
                    f"-input={dev_out}", f"-output={fb}"], check=True)
    host_o = os.path.join(work, "conv_igemm.o")
    subprocess.run(["hipcc", "-x", "hip"] + FLAGS + ["--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fb,
                                                     "-c", src, "-o", host_o], check=True)
    sys.path.insert(0, REPO)
    from vtoonify_amd import build
    objs = [host_o if s == "conv_igemm.hip" else os.path.join(build.HERE, "build", s.replace(".hip", ".o")) for s in build.SOURCES]
    lib = os.path.join(build.LIBDIR, f"libvtoonify_amd_{tag}.so")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
    print(lib)


if __name__ == "__main__":
    main()
