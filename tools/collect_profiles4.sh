#!/bin/bash
# Copies what is judged from a tools/profile_round4.sh run into profiles/:  tools/collect_profiles4.sh <tag> <prefix>
#   e.g. tools/collect_profiles4.sh r4_a r4_01   (gpurun_out/r4_a/* -> profiles/r4_01_*)
set -e
o=gpurun_out/$1; p=profiles/$2
for c in c2 c2b32 c4; do cp $o/stats_$c/k_kernel_stats.csv ${p}_${c}_kernel_stats.csv; done
cp $o/hbm_traffic.json profiles/hbm_traffic.json
cp $o/hbm_traffic.json ${p}_hbm_traffic.json
[ -f $o/pmc_c4_sq_summary.txt ] && cp $o/pmc_c4_sq_summary.txt ${p}_c4_sq_counters.txt
[ -f $o/kbench_gemm.txt ] && grep -v amdgpu.ids $o/kbench_gemm.txt > ${p}_kbench_gemm.txt
[ -f $o/kbench_attn_ln.txt ] && grep -v amdgpu.ids $o/kbench_attn_ln.txt > ${p}_kbench_attn_ln.txt
if [ -f $o/decode_bench.txt ]; then
  { echo "## tools/decode_bench.py"; grep -v amdgpu.ids $o/decode_bench.txt
    echo; echo "## tools/decode_trace.py (eager launches under rocprofv3 --kernel-trace)"; cat $o/decode_trace.txt; } > ${p}_decode.txt
fi
for f in bench_c2.json pytest_gpu_tail.txt bench_dp1.json; do [ -f $o/$f ] && cp $o/$f ${p}_$f; done
ls ${p}_*
