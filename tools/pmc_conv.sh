#!/bin/bash
# SQ counter passes over ONE conv of tools/conv_bench.py (round 4: what the patch kernels wait for).
#   gpurun -- 'bash tools/gpu.sh TAG sh "bash tools/pmc_conv.sh TAG \"=fus1 1024->512 @64\" VT_PATCH_PIPE 0 1"'
# prints, per kernel and per value of the switch, the mean counters per launch of two passes:
#   A  SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
#   B  SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL GRBM_GUI_ACTIVE
TAG=$1; ONLY=$2; VAR=$3; shift 3
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; export TMPDIR=/tmp
EXTRA=${PMC_CONV_ARGS:---stream --batch 4}
PA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
PB="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL GRBM_GUI_ACTIVE"
PC="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
for v in "$@"; do
  for pass in ${PMC_CONV_PASSES:-A B}; do   # C (PMC_CONV_PASSES="A B C"): instruction counts by class
    case $pass in A) C="$PA";; B) C="$PB";; *) C="$PC";; esac
    D=$O/pmcconv_${TAG}_${v}_$pass
    (cd /tmp && env $VAR=$v timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o p -- \
       python $R/tools/conv_bench.py $EXTRA --only "$ONLY" --iters 5 > $D.log 2>&1)
    echo "== $VAR=$v pass $pass"
    python - "$(find $D -name '*counter_collection.csv' | head -1)" <<'PY'
import collections, csv, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"]).split("(")[0][:60]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "conv" not in k: continue
    print(k, "launches", max(len(v) for v in d.values()))
    for c, v in d.items(): print(f"   {c:<28} {sum(v)/len(v):16.0f}")
PY
    rm -rf $D
  done
done
