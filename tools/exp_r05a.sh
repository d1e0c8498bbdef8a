#!/bin/bash
# Round 5, experiment A: the round-4 wrong-image-row defect of the lean epilogue with the fused ToRGB.
#   gpurun --timeout 900 -- 'bash tools/exp_r05a.sh'
# Arms (tools/flake_diag.py, D 2 x 64 x 96, two lanes in steady state, every wrong step explained from precomputed references):
#   exp1 = the round-4 form (SLP-packed ToRGB sums: `python -m vtoonify_amd.build --variant exp1 -- -fslp-vectorize`; at commit
#   af79b78, where these arms ran, it was -DVT_EXP=1), graph / eager / not-in-place image; exp3 = + system-scope skip loads (VT_EXP=3
#   at af79b78; gone since); product = the shipped form.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out; L=$GRAFT_REPO_ROOT/vtoonify_amd/lib
run() {  # tag lib steps mode [env...]
  tag=$1; lib=$2; steps=$3; mode=$4; shift 4
  ( env FLAKE_LIB=$L/$lib "$@" timeout 280 python tools/flake_diag.py D 2 64 96 $steps $mode $O/diag_$tag.json 2>&1 | grep -v amdgpu.ids ) > $O/diag_$tag.txt
  echo "== $tag: $(tail -1 $O/diag_$tag.txt)"
}
run exp1_graph libvtoonify_amd_exp1.so 800 graph
run exp1_eager libvtoonify_amd_exp1.so 400 eager
run exp1_split libvtoonify_amd_exp1.so 400 graph FLAKE_SPLIT=1
run exp3_graph libvtoonify_amd_exp3.so 400 graph
run product_graph libvtoonify_amd.so 600 graph
grep -h "best:" $O/diag_exp1_graph.txt | head -12
