#!/usr/bin/env python
"""Run bench.py against another build of the library (same-box A/B of the whole step):
    python tools/ab_step.py --lib gpurun_ab/libvt_<commit>.so -- --no-extras --no-video --no-cpu-baseline
The other build comes from `git worktree add /tmp/wt <commit>; (cd /tmp/wt; python -m vtoonify_amd.build)` and is copied to
gpurun_ab/ (git-ignored: *.so), so that it travels to the GPU box with the tree; the Python side is this tree's."""
import os
import runpy
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
args = sys.argv[1:]
lib = None
if "--lib" in args:
    i = args.index("--lib")
    lib = args[i + 1]
    del args[i:i + 2]
if "--" in args:
    args.remove("--")
if lib:
    from vtoonify_amd import _lib
    path = os.path.join(REPO, lib) if not os.path.isabs(lib) else lib
    _lib.DEFAULT_LIB = path      # (bench.py binds _lib.DEFAULT_LIB itself)
    _lib.use_library(path)
sys.argv = [os.path.join(REPO, "bench.py")] + args
runpy.run_path(sys.argv[0], run_name="__main__")
