# Same-box A/B of the whole 4-frame step: the library of commit 02d4381 (the END OF ROUND 5) against this tree's, three
# alternations.  The other build: git worktree add /tmp/wt 02d4381; (cd /tmp/wt; python -m vtoonify_amd.build); copy its
# vtoonify_amd/lib/libvtoonify_amd.so to gpurun_ab/libvt_02d4381.so (git-ignored, travels with the tree).
# -> profiles/r06_ab_round.txt
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for L in gpurun_ab/libvt_02d4381.so ""; do
    echo "## lib=${L:-this tree} rep $i"
    python tools/ab_step.py ${L:+--lib $L} -- --no-extras --no-video --no-cpu-baseline 2>/dev/null > /tmp/ab.json; python tools/bench_summary.py < /tmp/ab.json | head -2; python -c "import json;d=json.load(open(\"/tmp/ab.json\"));print(\"   \",[(k[\"kernel\"][:22],round(k[\"ms_per_step\"],3)) for k in d[\"kernels\"][:6]])"
  done
done
