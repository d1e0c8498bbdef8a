#!/bin/bash
# PMC passes over tools/conv_bench.py for one shape.  usage: tools_gpu_pmc.sh tag "only-substr" [extra conv_bench args]
TAG=$1; ONLY=$2; shift; shift
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # name, counters...
  local name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$name -o p -- python $GRAFT_REPO_ROOT/tools/conv_bench.py --only "$ONLY" --iters 3 $EXTRA > $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$name.log 2>&1)
}
EXTRA="$@"
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run tcp TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
for n in sq1 sq2 tcc tcp grbm; do echo "== $n"; tail -2 gpurun_out/pmc_${TAG}_$n.log | cut -c1-200; f=$(find gpurun_out/pmc_${TAG}_$n -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k=r.get('Kernel_Name','?')[:70]
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in agg.items():
    if 'conv' not in k: continue
    print(k)
    for c,v in d.items(): print(f"   {c:38s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
done
