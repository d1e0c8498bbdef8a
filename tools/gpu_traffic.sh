#!/bin/bash
# HBM traffic counters (separate --pmc passes, kernel-trace only) over the default bench command.
TAG=$1; shift
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph --op-iters 1 "$@" > $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$c.log 2>&1)
done
python tools/pmc_traffic.py $(find gpurun_out/pmc_${TAG}_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find gpurun_out/pmc_${TAG}_WRITE_SIZE -name "*counter_collection.csv" | head -1) > gpurun_out/pmc_traffic_$TAG.json
head -c 1500 gpurun_out/pmc_traffic_$TAG.json
