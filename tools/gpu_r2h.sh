#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 300 python -m pytest tests/test_ops.py -m gpu -q -x -k "transpose_blur" 2>&1 | tail -3
for v in 0 512; do for w in 256 512; do
VT_UPBLUR_PERSIST=$v VT_UPBLUR_WGS=$w timeout 60 python tools/conv_bench.py --only "up 64->32" --iters 50 --upblur 2>/dev/null | grep -v total | sed "s/^/PERSIST $v WGS $w /"
done; done
for ab in 11 12 14 13; do
VT_RGB_ABLATE=$ab timeout 60 python tools/conv_bench.py --only "up 64->32" --iters 50 --upblur 2>/dev/null | grep -v total | sed "s/^/ABLATE $ab /"
done
timeout 100 python tools/conv_bench.py --only "up " --iters 50 --upblur 2>/dev/null | grep -v total
for h in 0 100128064 100256064 200128064 200064064; do timeout 60 python tools/conv_bench.py --only "same 64 @512" --iters 30 --hint $h 2>/dev/null | grep -v total | sed "s/^/HINT $h /"; done
