#!/bin/bash
# GPU pass: parity tests, smoke, bench, rocprof stats.  usage: tools_gpu_run.sh [tag]
TAG=${1:-run}
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --maxfail=8 2>&1 | tail -40 > gpurun_out/pytest_gpu_$TAG.log
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1
echo "smoke rc=$?" >> gpurun_out/smoke_$TAG.log
timeout 600 python bench.py --steps 30 --warmup 3 --kernels > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
echo "bench rc=$?" >> gpurun_out/bench_$TAG.err
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1)
find gpurun_out/prof_$TAG -name "*.csv" | head; 
tail -5 gpurun_out/pytest_gpu_$TAG.log; tail -3 gpurun_out/smoke_$TAG.log; tail -60 gpurun_out/bench_$TAG.err
