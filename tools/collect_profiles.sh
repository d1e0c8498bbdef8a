#!/bin/bash
# Copy the artifacts of one full GPU pass (tools/gpu.sh TAG smoke tests bench prof pmc op stress) from the scratch directory
# gpurun_out/ into the tracked profiles/ under this round's names.    usage: tools/collect_profiles.sh r06s [r06]
set -e
tag=$1; rnd=${2:-r06}
cd "$(dirname "$0")/.."
cpy() { [ -s "gpurun_out/$1" ] && cp "gpurun_out/$1" "profiles/$2" && echo "profiles/$2 <- gpurun_out/$1" || echo "missing gpurun_out/$1"; }
cpy bench_$tag.json ${rnd}_bench.json
cpy bench_${tag}_kernels.txt ${rnd}_bench_kernels.txt
cpy rocprofv3_kernel_stats_lanes1_$tag.txt ${rnd}_rocprofv3_kernel_stats_lanes1.txt
cpy rocprofv3_kernel_stats_lanes3_$tag.txt ${rnd}_rocprofv3_kernel_stats_lanes3.txt
cpy pmc_traffic_$tag.json ${rnd}_pmc_traffic.json
cpy pmc_mfma_$tag.json ${rnd}_pmc_mfma.json
cpy op_bench_bf16_$tag.json ${rnd}_op_bench_bf16.json
cpy op_bench_fp32_$tag.json ${rnd}_op_bench_fp32.json
cpy parity_metrics_$tag.jsonl ${rnd}_parity_metrics.jsonl
cpy pytest_gpu_$tag.log ${rnd}_pytest_gpu.log
cpy smoke_$tag.log ${rnd}_smoke.log
cpy stress_$tag.txt ${rnd}_stress.txt
