#!/bin/bash
# SQ counter passes over ONE stand-alone operator call (round 6: what upfirdn2d's bf16 blur waits for).
#   gpurun -- 'bash tools/gpu.sh TAG sh "bash tools/pmc_op.sh TAG VAR v1 v2 ..."'      (VAR: an environment switch the library reads, or any unused name with one value)
TAG=$1; VAR=$2; shift 2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; export TMPDIR=/tmp
PA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
PB="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL GRBM_GUI_ACTIVE"
PC="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"
cat > /tmp/one_op.py <<'PY'
import sys, torch
sys.path.insert(0, sys.argv[1])
from vtoonify_amd import _lib, synth
from vtoonify_amd.op import upfirdn2d
_lib.use_library(_lib.DEFAULT_LIB)
dev = torch.device("cuda:0")
k = synth.fir_kernel_2d().to(dev)
x = torch.randn(1, 32, 1025, 1025, device=dev).to(torch.bfloat16)
for _ in range(5):
    y = upfirdn2d(x, k * 4, pad=(1, 1))
torch.cuda.synchronize()
PY
for v in "$@"; do
  for pass in A B C; do
    case $pass in A) C="$PA";; B) C="$PB";; *) C="$PC";; esac
    D=$O/pmcop_${TAG}_${v}_$pass
    (cd /tmp && env $VAR=$v timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o p -- python /tmp/one_op.py $R > $D.log 2>&1)
    echo "== $VAR=$v pass $pass"
    python - "$(find $D -name '*counter_collection.csv' | head -1)" <<'PY'
import collections, csv, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"]).split("(")[0][:60]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "upfirdn" not in k: continue
    print(k, "launches", max(len(v) for v in d.values()))
    for c, v in d.items(): print(f"   {c:<28} {sum(v)/len(v):16.0f}")
PY
    rm -rf $D
  done
done
