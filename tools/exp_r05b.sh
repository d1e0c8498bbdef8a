#!/bin/bash
# Round 5, experiment B: is the wrong-image-row defect a write-after-read hazard of packed fp32 instructions?
#   gpurun --timeout 1200 -- 'bash tools/exp_r05b.sh'
#   1. tools/probe/bin/pk_war_probe: the instruction pair alone (v_pk_add_f32 ; VALU write of its source) under aggressor kernels
#   2. the failing library with `s_nop 1` patched into its ISA AFTER (warA) / BEFORE (warB, control) every packed-fp32
#      instruction whose source the next VALU instruction overwrites (tools/patch_isa_build.py), tools/flake_diag.py each
#   3. the product against the same sources built with -fno-slp-vectorize (libvtoonify_amd_noslp.so): two-lane diag + bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out; L=$GRAFT_REPO_ROOT/vtoonify_amd/lib
( timeout 200 tools/probe/bin/pk_war_probe 2 ) > $O/pk_war_probe.txt 2>&1; echo "probe rc=$?" >> $O/pk_war_probe.txt
grep -v "lo errors          0  hi errors          0" $O/pk_war_probe.txt | head -40
run() {  # tag lib steps mode [env...]
  tag=$1; lib=$2; steps=$3; mode=$4; shift 4
  ( env FLAKE_LIB=$L/$lib FLAKE_MAXDIAG=20 "$@" timeout 280 python tools/flake_diag.py D 2 64 96 $steps $mode $O/diag_$tag.json 2>&1 | grep -v amdgpu.ids ) > $O/diag_$tag.txt
  echo "== $tag: $(tail -1 $O/diag_$tag.txt)"
}
run warA libvtoonify_amd_warA.so 800 graph
run warB libvtoonify_amd_warB.so 400 graph
if [ -f $L/libvtoonify_amd_noslp.so ]; then
  run noslp libvtoonify_amd_noslp.so 400 graph
  B="--steps 30 --warmup 5 --no-extras --no-video --no-cpu-baseline --kernels"
  for rep in 1 2; do
    timeout 200 python bench.py $B > $O/benchq_prod_$rep.json 2> $O/benchq_prod_$rep.err
    cp $L/libvtoonify_amd.so /tmp/prod.so; cp $L/libvtoonify_amd_noslp.so $L/libvtoonify_amd.so
    timeout 200 python bench.py $B > $O/benchq_noslp_$rep.json 2> $O/benchq_noslp_$rep.err
    cp /tmp/prod.so $L/libvtoonify_amd.so
  done
  python - <<'EOF'
import json
for t in ("prod_1", "noslp_1", "prod_2", "noslp_2"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/benchq_{t}.json") if l.startswith("{")][-1])
        print(t, round(d["value"], 1), "frames/s", "single", round(d["single_stream"]["value"], 1), "dominant", d["roofline"]["kernel"], round(d["roofline"]["avg_launch_us"], 1), "us frac", round(d["roofline"]["frac"], 3), "kernel sum", round(d["roofline"]["kernel_sum_ms_per_frame"], 3))
    except Exception as e:
        print(t, "failed", e)
EOF
fi
