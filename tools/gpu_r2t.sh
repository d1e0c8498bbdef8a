#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_ops.py -m gpu -q -x -k "whole_k or transpose_blur" 2>&1 | tail -1
CB="python tools/conv_bench.py --iters 100 --upblur"
for rep in 1 2; do
for S in 0 1 2 4 8; do echo "skew $S"; VT_UPBLUR_SKEW=$S timeout 90 $CB --only "up " 2>&1 | grep -v "^total\|amdgpu" | awk '{print $1,$2,$3,$(NF-5),$(NF-4)}'; done
done
echo "adain chain (fullk with in_tile_stats)"; timeout 60 python tools/conv_bench.py --stream --hint 400000000 --iters 200 --only "=res 512->512 @32" 2>&1 | grep -v "^total\|amdgpu"
