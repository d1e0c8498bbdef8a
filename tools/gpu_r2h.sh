#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
for db in 4 99; do
VT_UPBLUR_DB=$db timeout 60 python tools/conv_bench.py --only "up " --iters 50 --upblur 2>/dev/null | grep -v total | awk -v t="DBMIN=$db" '{print t, $1,$2,$3, $(NF-5), $(NF-4)}'
VT_UPBLUR_DB=$db timeout 60 python tools/conv_bench.py --only "up " --iters 50 --upblur --hint 16 2>/dev/null | grep -v total | awk -v t="DBMIN=$db CN16" '{print t, $1,$2,$3, $(NF-5), $(NF-4)}'
done
done
