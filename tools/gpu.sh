#!/bin/bash
# One parameterised GPU pass for gpurun (replaces the per-experiment gpu_r2?.sh scripts of round 2).
#   gpurun --timeout 900 -- 'bash tools/gpu.sh TAG stage [stage ...]'
# stages (run in the order given; every stage writes under gpurun_out/ with the tag in the file name):
#   probe        tools/probe/bin/glds_probe2 (LDS-DMA base alignment / EXEC mask / 160 KiB static LDS)
#   tests        pytest -m gpu (+ parity metrics)
#   testsq K     pytest -m gpu -k K
#   bench        bench.py with default flags + per-kernel table
#   benchq       bench.py --no-extras --no-video --no-cpu-baseline (value, single_stream, kernel table)
#   prof         rocprofv3 --kernel-trace --stats of the bench command, one and three frames in flight
#   prof1        the lanes-1 half of prof
#   pmc          rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, MFMA busy) over the eager single-stream command
#   smoke        __graft_entry__.smoke()
#   stress       two engine lanes / the video driver in steady state against serial results (tools/flake_lanes.py, flake_video.py)
#   conv ARGS    tools/conv_bench.py ARGS   (quote ARGS as one word, e.g. "--stream --only res --batch 4")
#   op           tools/op_bench.py bf16 + fp32
#   sh CMD       arbitrary command (one word), e.g. "bash tools/ab.sh adain" (the A/B experiments of round 3)
TAG=${1:-r04}; shift
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
B="python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-video --no-extras --min-seconds 0.3"
P="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-video --no-extras --min-seconds 0 --lanes 1 --no-graph --op-iters 1"
n=0
while [ $# -gt 0 ]; do
  st=$1; shift; n=$((n+1))
  case $st in
    probe) timeout 60 tools/probe/bin/glds_probe2 > $O/probe_$TAG.txt 2>&1; cat $O/probe_$TAG.txt ;;
    tests)
      rm -f $O/parity_metrics.jsonl
      ( time timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) > $O/pytest_gpu_$TAG.log 2>&1; tail -6 $O/pytest_gpu_$TAG.log
      cp $O/parity_metrics.jsonl $O/parity_metrics_$TAG.jsonl 2>/dev/null ;;
    testsq) K=$1; shift
      ( time timeout 900 python -m pytest tests -m gpu -q -x -k "$K" 2>&1 | tail -25 ) > $O/pytest_gpu_${TAG}_$n.log 2>&1; tail -8 $O/pytest_gpu_${TAG}_$n.log ;;
    bench)
      timeout 600 python bench.py --kernels > $O/bench_$TAG.json 2> $O/bench_$TAG.err; echo "bench rc=$?" >> $O/bench_$TAG.err
      grep -v "^W\|^E\|amdgpu.ids" $O/bench_$TAG.err > $O/bench_${TAG}_kernels.txt
      grep '"metric"' $O/bench_$TAG.json | python tools/bench_summary.py ; tail -1 $O/bench_$TAG.err ;;
    benchq)
      timeout 300 python bench.py --kernels --no-extras --no-video --no-cpu-baseline > $O/benchq_$TAG.json 2> $O/benchq_$TAG.err; echo "bench rc=$?" >> $O/benchq_$TAG.err
      grep -v "^W\|^E\|amdgpu.ids" $O/benchq_$TAG.err > $O/benchq_${TAG}_kernels.txt
      grep '"metric"' $O/benchq_$TAG.json | python tools/bench_summary.py ; head -24 $O/benchq_${TAG}_kernels.txt ;;
    prof)
      for L in 1 3; do
        (cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof${L}_$TAG -o bench -- $B --lanes $L > $GRAFT_REPO_ROOT/$O/prof${L}_$TAG.log 2>&1)
        python tools/rocpd_stats.py $(find $O/prof${L}_$TAG -name "*.db" | head -1) > $O/rocprofv3_kernel_stats_lanes${L}_$TAG.txt 2>&1
        rm -rf $O/prof${L}_$TAG
      done
      head -14 $O/rocprofv3_kernel_stats_lanes1_$TAG.txt | cut -c1-150 ;;
    prof1)   # lanes 1 only (short calls)
      (cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof1_$TAG -o bench -- $B --lanes 1 > $GRAFT_REPO_ROOT/$O/prof1_$TAG.log 2>&1)
      python tools/rocpd_stats.py $(find $O/prof1_$TAG -name "*.db" | head -1) > $O/rocprofv3_kernel_stats_lanes1_$TAG.txt 2>&1
      rm -rf $O/prof1_$TAG
      head -14 $O/rocprofv3_kernel_stats_lanes1_$TAG.txt | cut -c1-150 ;;
    pmc)
      for c in FETCH_SIZE WRITE_SIZE; do
        (cd /tmp && timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_${TAG}_$c -o p -- $P > $GRAFT_REPO_ROOT/$O/pmc_${TAG}_$c.log 2>&1)
      done
      python tools/pmc_traffic.py $(find $O/pmc_${TAG}_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc_${TAG}_WRITE_SIZE -name "*counter_collection.csv" | head -1) > $O/pmc_traffic_$TAG.json 2> $O/pmc_traffic_$TAG.err
      rm -rf $O/pmc_${TAG}_FETCH_SIZE $O/pmc_${TAG}_WRITE_SIZE
      (cd /tmp && timeout 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_${TAG}_mfma -o p -- $P > $GRAFT_REPO_ROOT/$O/pmc_${TAG}_mfma.log 2>&1)
      python tools/pmc_mfma.py $(find $O/pmc_${TAG}_mfma -name "*counter_collection.csv" | head -1) > $O/pmc_mfma_$TAG.json 2> $O/pmc_mfma_$TAG.err
      rm -rf $O/pmc_${TAG}_mfma
      python -c "
import json; d=json.load(open('$O/pmc_mfma_$TAG.json'))['kernels']
for k,v in sorted(d.items(), key=lambda kv:-kv[1].get('sq_valu_mfma_busy_cycles_per_launch',0))[:8]: print(k[:70], round(v.get('mfma_utilisation',0),3))
t=json.load(open('$O/pmc_traffic_$TAG.json'))['kernels']
for k,v in sorted(t.items(), key=lambda kv:-kv[1]['hbm_bytes_per_launch']*kv[1]['launches_sampled'])[:8]: print(k[:70], round(v['hbm_bytes_per_launch']/1e6,2),'MB/launch', v['launches_sampled'])
" ;;
    stress)
      ( for a in "D 2 64 96 400" "D 4 256 256 100" "T 2 64 96 300" "D 3 72 104 200"; do timeout 300 python tools/flake_lanes.py $a graph 2>&1 | tail -1; done
        timeout 300 python tools/flake_video.py 20 only22 2>&1 | tail -1 ) > $O/stress_$TAG.txt 2>&1; cat $O/stress_$TAG.txt ;;
    smoke) timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_$TAG.log 2>&1; echo "smoke rc=$?" >> $O/smoke_$TAG.log; tail -3 $O/smoke_$TAG.log ;;
    conv) A=$1; shift
      timeout 300 python tools/conv_bench.py $A > $O/conv_${TAG}_$n.txt 2>&1; echo "# conv_bench $A" >> $O/conv_${TAG}_$n.txt; cat $O/conv_${TAG}_$n.txt | grep -v amdgpu.ids ;;
    op)
      for d in bf16 fp32; do timeout 200 python tools/op_bench.py --dtype $d --json $O/op_bench_${d}_$TAG.json > $O/op_bench_${d}_$TAG.txt 2>&1; cat $O/op_bench_${d}_$TAG.txt | grep -v amdgpu.ids; done ;;
    sh) A=$1; shift
      timeout 600 bash -c "$A" > $O/sh_${TAG}_$n.txt 2>&1; echo "rc=$?" >> $O/sh_${TAG}_$n.txt; tail -40 $O/sh_${TAG}_$n.txt ;;
    *) echo "unknown stage $st" ;;
  esac
done
