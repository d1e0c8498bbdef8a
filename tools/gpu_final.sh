#!/bin/bash
# One prioritized GPU pass: bench line, rocprofv3 kernel stats, PMC traffic, smoke, parity tests.
# usage: tools/gpu_final.sh tag [pytest-seconds]
TAG=${1:-f}; PYT=${2:-240}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 400 python bench.py --kernels > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?" >> gpurun_out/bench_$TAG.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-video --lanes 1 > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1)
python tools/rocpd_stats.py $(find gpurun_out/prof_$TAG -name "*.db" | head -1) > gpurun_out/prof_${TAG}_stats.txt 2>&1
rm -rf gpurun_out/prof_$TAG
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof3_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-video --lanes 3 > $GRAFT_REPO_ROOT/gpurun_out/prof3_$TAG.log 2>&1)
python tools/rocpd_stats.py $(find gpurun_out/prof3_$TAG -name "*.db" | head -1) > gpurun_out/prof3_${TAG}_stats.txt 2>&1
rm -rf gpurun_out/prof3_$TAG
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-video --lanes 1 --no-graph --op-iters 1 > $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$c.log 2>&1)
done
python tools/pmc_traffic.py $(find gpurun_out/pmc_${TAG}_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find gpurun_out/pmc_${TAG}_WRITE_SIZE -name "*counter_collection.csv" | head -1) > gpurun_out/pmc_traffic_$TAG.json 2> gpurun_out/pmc_traffic_$TAG.err
rm -rf gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke_$TAG.log
timeout $PYT python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/pytest_gpu_$TAG.log
grep '"metric"' gpurun_out/bench_$TAG.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'])"
tail -3 gpurun_out/smoke_$TAG.log; tail -4 gpurun_out/pytest_gpu_$TAG.log; cut -c1-140 gpurun_out/prof_${TAG}_stats.txt | head -14
