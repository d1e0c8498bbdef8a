#!/bin/bash
# first GPU pass: parity tests, smoke, bench, rocprof stats
set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --maxfail=8 -x 2>&1 | tail -60 > gpurun_out/pytest_gpu.log
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 600 python bench.py --steps 20 --warmup 3 --kernels > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "bench rc=$?" >> gpurun_out/bench.err
tail -5 gpurun_out/pytest_gpu.log; tail -3 gpurun_out/smoke.log; cat gpurun_out/bench.json; tail -80 gpurun_out/bench.err
