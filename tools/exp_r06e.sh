# Same-box A/B of the whole 4-frame step: the library of commit 9cbeb86 (the round's state before the flat up-sampling tiles, the
# XCD-contiguous thin convs, the parallel row fold and the c32 ring) against this tree's, three alternations.
# -> profiles/r06_ab_step.txt
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for L in gpurun_ab/libvt_9cbeb86.so ""; do
    echo "## lib=${L:-this tree} rep $i"
    python tools/ab_step.py ${L:+--lib $L} -- --no-extras --no-video --no-cpu-baseline 2>/dev/null > /tmp/ab.json; python tools/bench_summary.py < /tmp/ab.json | head -2; python -c "import json;d=json.load(open(\"/tmp/ab.json\"));print(\"   \",[(k[\"kernel\"][:22],round(k[\"ms_per_step\"],3)) for k in d[\"kernels\"][:6]])"
  done
done
