#!/bin/bash
# One prioritized GPU pass: bench line, rocprofv3 kernel stats (one frame in flight, then the default
# three), PMC traffic, engine parity tests, smoke.  usage: tools/gpu_final.sh tag
TAG=${1:-f}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-video"
timeout 120 python bench.py --kernels > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?" >> gpurun_out/bench_$TAG.err
(cd /tmp && timeout 60 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o bench -- $B --lanes 1 > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1)
python tools/rocpd_stats.py $(find gpurun_out/prof_$TAG -name "*.db" | head -1) > gpurun_out/prof_${TAG}_stats.txt 2>&1
rm -rf gpurun_out/prof_$TAG
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 50 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-video --lanes 1 --no-graph --op-iters 1 > $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$c.log 2>&1)
done
python tools/pmc_traffic.py $(find gpurun_out/pmc_${TAG}_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find gpurun_out/pmc_${TAG}_WRITE_SIZE -name "*counter_collection.csv" | head -1) > gpurun_out/pmc_traffic_$TAG.json 2> gpurun_out/pmc_traffic_$TAG.err
rm -rf gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE
(cd /tmp && timeout 50 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_mfma -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-video --lanes 1 --no-graph --op-iters 1 > $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_mfma.log 2>&1)
python tools/pmc_mfma.py $(find gpurun_out/pmc_${TAG}_mfma -name "*counter_collection.csv" | head -1) > gpurun_out/pmc_mfma_$TAG.json 2> gpurun_out/pmc_mfma_$TAG.err
rm -rf gpurun_out/pmc_${TAG}_mfma
timeout 60 python -m pytest tests/test_engine.py tests/test_bisenet.py -m gpu -q -x -k "golden or flight" 2>&1 | tail -4 > gpurun_out/pytest_gpu_$TAG.log
(cd /tmp && timeout 60 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof3_$TAG -o bench -- $B --lanes 3 > $GRAFT_REPO_ROOT/gpurun_out/prof3_$TAG.log 2>&1)
python tools/rocpd_stats.py $(find gpurun_out/prof3_$TAG -name "*.db" | head -1) > gpurun_out/prof3_${TAG}_stats.txt 2>&1
rm -rf gpurun_out/prof3_$TAG
timeout 40 python tools/bisenet_bench.py --no-cpu --steps 50 2>&1 | grep -v "^W\|^E\|amdgpu.ids" > gpurun_out/bisenet_bench_$TAG.txt
timeout 40 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke_$TAG.log
grep '"metric"' gpurun_out/bench_$TAG.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['single_stream'], d['roofline'])"
tail -3 gpurun_out/smoke_$TAG.log; tail -2 gpurun_out/pytest_gpu_$TAG.log; cut -c1-140 gpurun_out/prof_${TAG}_stats.txt | head -8
