#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
for rep in 1 2; do
for ab in 0 11 12 14 16 13; do
  for only in "up 64->32"; do
    VT_RGB_ABLATE=$ab timeout 60 python tools/conv_bench.py --only "$only" --iters 50 --upblur 2>/dev/null | grep -v total | sed "s/^/ABLATE $ab /"
  done
done
done > $O/upblur_ablate.txt 2>&1
cat $O/upblur_ablate.txt
(cd /tmp && timeout 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_ub -o p -- python $GRAFT_REPO_ROOT/tools/conv_bench.py --only "up 64->32" --iters 5 --upblur > $GRAFT_REPO_ROOT/$O/pmc_ub.log 2>&1)
(cd /tmp && timeout 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_ub2 -o p -- python $GRAFT_REPO_ROOT/tools/conv_bench.py --only "up 64->32" --iters 5 --upblur > $GRAFT_REPO_ROOT/$O/pmc_ub2.log 2>&1)
python - <<'PY'
import csv,glob,collections
for dn in ('pmc_ub','pmc_ub2'):
    f=glob.glob(f'gpurun_out/{dn}/**/*counter_collection.csv',recursive=True)
    if not f: print(dn,'no csv'); continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if 'upblur' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print({c: round(sum(v)/len(v)) for c,v in acc.items()})
PY
tail -3 $O/pmc_ub2.log
rm -rf $O/pmc_ub $O/pmc_ub2
