#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_ops.py -m gpu -q -x -k "c64 or fused_torgb" 2>&1 | tail -1
VT_C64_PIPE=1 timeout 200 python -m pytest tests/test_ops.py -m gpu -q -x -k "c64 or fused_torgb" 2>&1 | tail -1
VT_C64_ROWS=16 timeout 200 python -m pytest tests/test_ops.py -m gpu -q -x -k "c64 or fused_torgb" 2>&1 | tail -1
CB="python tools/conv_bench.py --iters 100"
for rep in 1 2; do
for cfg in "VT_C64_ROWS=16" "VT_C64_ROWS=8" "VT_C64_PIPE=1" "VT_C64_KERNEL=0"; do
echo "$cfg"; env $cfg timeout 60 $CB --only "same 64 @512" 2>&1 | grep -v "^total\|amdgpu"; env $cfg timeout 60 $CB --only "same 64 @512" --rgb 2>&1 | grep -v "^total\|amdgpu"
done
for m in 31 32 33; do echo "rows8 ablate $m"; VT_RGB_ABLATE=$m timeout 60 $CB --only "same 64 @512" 2>&1 | grep -v "^total\|amdgpu"; done
done
