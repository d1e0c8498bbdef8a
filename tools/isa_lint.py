#!/usr/bin/env python
"""ISA lint of the built gfx950 objects (vtoonify_amd/build/*.o): what the source cannot show and hipcc does not check.

    python tools/isa_lint.py [--packed] [--obj conv_igemm] [substring ...]

Three scans per kernel (tests/test_isa_lint.py runs them on the product build, CPU only -- hipcc cross-compiles):

* hidden loads -- the kernels keep vector-memory operations in flight across barriers and wait for them with hand-counted
  `s_waitcnt vmcnt(N)`; the register-destination ones are inline asm the compiler does not model (vt_common.hpp:
  vt_gload16_pair_hidden, vt_bload_hidden).  A destination register touched while its load can still be outstanding is the
  race behind round 4's NaN tiles.  The scan replays every straight-line stretch of a kernel: a load's
  destination registers stay "pending" until a vmcnt wait retires it (vector memory retires in issue order); any other
  instruction that names a pending register is a violation.  Branch targets drop what they cannot know (conservative:
  nothing pending is assumed at a label, so the scan under-reports across branches, never over-reports).
* packed fp32 -- `v_pk_{mul,add,fma}_f32` per kernel, how many carry op_sel / op_sel_hi lane selects, and how many of their
  results are consumed by the very next instruction (the shape round 4 blamed for a wrong image row).
* scratch inside the matrix loop -- no scratch_/buffer scratch access between the first and the last MFMA of a kernel in
  layout order (a spill reload shares vmcnt with the counted LDS-DMA and would drain it every step).
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
_REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
_KERNEL = re.compile(r"^[0-9a-f]+ <(.+)>:$")


def disassemble(obj_path):
    """{mangled kernel symbol: [instruction text, ...]} of the gfx950 code object inside a host object file"""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat"), os.path.join(tmp, "co")
        r = subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj_path], capture_output=True)
        if r.returncode != 0 or not os.path.exists(fat):
            return out   # (an object without device code)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True, capture_output=True)
        txt = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True,
                             text=True).stdout
    cur = None
    for line in txt.split("\n"):
        m = _KERNEL.match(line)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        if cur is None:
            continue
        t = line.split("//")[0].strip()
        if t:
            cur.append(t)
    return out


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return dict(zip(names, r.stdout.split("\n")))


def _regs(tok_text):
    s = set()
    for m in _REG.finditer(tok_text):
        if m.group(3) is not None:
            s.add(int(m.group(3)))
        else:
            s.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return s


def _operands(t):
    parts = t.split(None, 1)
    return parts[0], (parts[1] if len(parts) > 1 else "")


_BRANCH = ("s_cbranch", "s_branch", "s_setpc", "s_endpgm", "s_call")


def scan_hidden_loads(lines):
    """[(index, instruction, index of the pending load)]"""
    pend = []   # (destination registers, index) in issue order; LDS-DMA loads and stores carry an empty set
    bad = []
    # objdump prints branch targets as offsets, not labels: every instruction after a branch may be a target.  Treat a
    # branch as the end of a stretch.
    for i, t in enumerate(lines):
        op, rest = _operands(t)
        if op.startswith(_BRANCH):
            pend = []
            continue
        is_load = op.startswith(("global_load", "buffer_load", "scratch_load", "flat_load"))
        is_store = op.startswith(("global_store", "buffer_store", "scratch_store", "flat_store", "global_atomic", "buffer_atomic"))
        if is_load or is_store:
            toks = [x.strip() for x in rest.split(",")]
            lds = " lds" in t
            dst = _regs(toks[0]) if (is_load and not lds and toks) else set()
            srcs = _regs(",".join(toks[1:])) if is_load and not lds else _regs(rest)
            for r, ln in pend:
                if r & srcs:
                    bad.append((i, t, ln))
            pend.append((dst, i))
            continue
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", t)
            if m:
                n = int(m.group(1))
                pend = pend[len(pend) - n:] if n < len(pend) else pend
                if n == 0:
                    pend = []
            continue
        touched = _regs(rest)
        if touched:
            for r, ln in pend:
                if r & touched:
                    bad.append((i, t, ln))
    return bad


def packed_f32(lines):
    """{"n": v_pk_*_f32 count, "sel": with op_sel / op_sel_hi, "back_to_back": result read by the next instruction,
        "b2b_mem": ... by a DS / vector-memory instruction}"""
    n = sel = b2b = b2b_mem = 0
    for i, t in enumerate(lines):
        op, rest = _operands(t)
        if not re.match(r"v_pk_(mul|add|fma)_f32", op):
            continue
        n += 1
        if "op_sel" in t:
            sel += 1
        dst = _regs(rest.split(",")[0])
        j = i + 1
        while j < len(lines) and lines[j].startswith(("s_nop", "s_waitcnt")):
            if lines[j].startswith("s_nop"):
                j = len(lines)   # a wait state was inserted: not back to back
                break
            j += 1
        if j < len(lines):
            op2, rest2 = _operands(lines[j])
            toks2 = rest2.split(",")
            srcs2 = _regs(",".join(toks2[1:])) if not op2.startswith(("ds_write", "global_store", "buffer_store")) else _regs(rest2)
            if dst & srcs2:
                b2b += 1
                if op2.startswith(("ds_", "global_", "buffer_", "flat_", "scratch_")):
                    b2b_mem += 1
    return {"n": n, "sel": sel, "back_to_back": b2b, "b2b_mem": b2b_mem}


def scratch_inside_mfma(lines):
    """scratch accesses between the first and the last MFMA (layout order)"""
    mf = [i for i, t in enumerate(lines) if t.startswith("v_mfma") or t.startswith("v_smfma")]
    if not mf:
        return []
    return [(i, t) for i, t in enumerate(lines[mf[0]:mf[-1] + 1], mf[0]) if t.startswith("scratch_")]


def objects(obj_dir=None):
    obj_dir = obj_dir or os.path.join(REPO, "vtoonify_amd", "build")
    return sorted(os.path.join(obj_dir, f) for f in os.listdir(obj_dir) if f.endswith(".o"))


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    only_obj = None
    if "--obj" in sys.argv:
        only_obj = sys.argv[sys.argv.index("--obj") + 1]
        args = [a for a in args if a != only_obj]
    total_bad = 0
    rows = []
    for o in objects():
        if only_obj and os.path.basename(o) != only_obj + ".o":
            continue
        kern = disassemble(o)
        names = demangle(list(kern))
        for k, lines in kern.items():
            dn = names.get(k, k)
            if args and not any(a in dn for a in args):
                continue
            bad = scan_hidden_loads(lines)
            sc = scratch_inside_mfma(lines)
            pk = packed_f32(lines)
            total_bad += len(bad) + len(sc)
            for i, t, ln in bad[:5]:
                print(f"{dn[:90]}: [{i}] {t}   touches registers of the load at [{ln}] {lines[ln]}")
            for i, t in sc[:5]:
                print(f"{dn[:90]}: [{i}] {t}   scratch access inside the matrix loop")
            rows.append((dn, len(lines), pk))
    if "--packed" in sys.argv:
        print(f"{'kernel':<100} {'insts':>7} {'pk_f32':>7} {'op_sel':>7} {'b2b':>5} {'b2b_mem':>7}")
        for dn, n, pk in sorted(rows, key=lambda r: -r[2]["n"]):
            if pk["n"]:
                short = re.sub(r"\(anonymous namespace\)::", "", dn)
                short = re.sub(r"\(.*$", "", short)
                print(f"{short[:100]:<100} {n:7d} {pk['n']:7d} {pk['sel']:7d} {pk['back_to_back']:5d} {pk['b2b_mem']:7d}")
    print("kernels:", len(rows), "violations:", total_bad)
