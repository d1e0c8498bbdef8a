#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
for ab in 0 11 12 13; do
  for only in "up 64->32" "up 256->128"; do
    VT_RGB_ABLATE=$ab timeout 60 python tools/conv_bench.py --only "$only" --iters 50 --upblur 2>/dev/null | grep -v total | sed "s/^/ABLATE $ab /"
  done
done > $O/upblur_ablate.txt 2>&1
cat $O/upblur_ablate.txt
(cd /tmp && timeout 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_ub -o p -- python $GRAFT_REPO_ROOT/tools/conv_bench.py --only "up 64->32" --iters 5 --upblur > $GRAFT_REPO_ROOT/$O/pmc_ub.log 2>&1)
python - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/pmc_ub/**/*counter_collection.csv',recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    if 'upblur' in r['Kernel_Name']: acc['upblur'][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in acc.items():
    print(k,{c: sum(v)/len(v) for c,v in d.items()})
PY
rm -rf $O/pmc_ub
