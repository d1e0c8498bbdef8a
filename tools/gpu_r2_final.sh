#!/bin/bash
# Round-2 deliverables pass: bench line (default flags), rocprofv3 kernel stats with one and three frames in flight,
# PMC passes (HBM traffic, MFMA utilisation) over the same command, parity metrics, smoke.  usage: tools/gpu_r2_final.sh tag [tests]
TAG=${1:-r02}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
B="python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-video --no-extras --min-seconds 0.3"
if [ "$2" = "tests" ]; then
  rm -f $O/parity_metrics.jsonl
  ( time timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > $O/pytest_gpu_$TAG.log 2>&1; tail -5 $O/pytest_gpu_$TAG.log
  cp $O/parity_metrics.jsonl $O/parity_metrics_$TAG.jsonl 2>/dev/null
fi
timeout 300 python bench.py --kernels > $O/bench_$TAG.json 2> $O/bench_$TAG.err; echo "bench rc=$?" >> $O/bench_$TAG.err
grep -v "^W\|^E\|amdgpu.ids" $O/bench_$TAG.err > $O/bench_${TAG}_kernels.txt
for L in 1 3; do
  (cd /tmp && timeout 90 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof${L}_$TAG -o bench -- $B --lanes $L > $GRAFT_REPO_ROOT/$O/prof${L}_$TAG.log 2>&1)
  python tools/rocpd_stats.py $(find $O/prof${L}_$TAG -name "*.db" | head -1) > $O/rocprofv3_kernel_stats_lanes${L}_$TAG.txt 2>&1
  rm -rf $O/prof${L}_$TAG
done
P="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-video --no-extras --min-seconds 0 --lanes 1 --no-graph --op-iters 1"
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 90 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_${TAG}_$c -o p -- $P > $GRAFT_REPO_ROOT/$O/pmc_${TAG}_$c.log 2>&1)
done
python tools/pmc_traffic.py $(find $O/pmc_${TAG}_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc_${TAG}_WRITE_SIZE -name "*counter_collection.csv" | head -1) > $O/pmc_traffic_$TAG.json 2> $O/pmc_traffic_$TAG.err
rm -rf $O/pmc_${TAG}_FETCH_SIZE $O/pmc_${TAG}_WRITE_SIZE
(cd /tmp && timeout 90 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_${TAG}_mfma -o p -- $P > $GRAFT_REPO_ROOT/$O/pmc_${TAG}_mfma.log 2>&1)
python tools/pmc_mfma.py $(find $O/pmc_${TAG}_mfma -name "*counter_collection.csv" | head -1) > $O/pmc_mfma_$TAG.json 2> $O/pmc_mfma_$TAG.err
rm -rf $O/pmc_${TAG}_mfma
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_$TAG.log 2>&1; echo "smoke rc=$?" >> $O/smoke_$TAG.log
grep '"metric"' $O/bench_$TAG.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['single_stream']['value'], d.get('module_call',{}).get('value'), d.get('batch4',{}).get('value'), d.get('config3',{}).get('value'), d['roofline'], d.get('pcie_inclusive'), d.get('cpu_baseline'))"
tail -2 $O/smoke_$TAG.log; head -12 $O/rocprofv3_kernel_stats_lanes1_$TAG.txt | cut -c1-150
python -c "
import json; d=json.load(open('$O/pmc_mfma_$TAG.json'))['kernels']
for k,v in sorted(d.items(), key=lambda kv:-kv[1].get('sq_valu_mfma_busy_cycles_per_launch',0))[:8]: print(k[:70], round(v.get('mfma_utilisation',0),3))
"
