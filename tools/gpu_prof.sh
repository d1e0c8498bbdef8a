#!/bin/bash
# rocprofv3 kernel trace of the default bench command.  usage: tools_gpu_prof.sh tag [bench args]
TAG=$1; shift
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 3 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1)
python tools/rocpd_stats.py $(find gpurun_out/prof_$TAG -name "*.db" | head -1) > gpurun_out/prof_${TAG}_stats.txt
cut -c1-150 gpurun_out/prof_${TAG}_stats.txt | head -32
grep '"metric"' gpurun_out/prof_$TAG.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
