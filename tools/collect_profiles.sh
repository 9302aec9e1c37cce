#!/bin/bash
# Copies what is judged from a tools/profile_round3.sh run into profiles/:  tools/collect_profiles.sh <tag> <prefix>
#   e.g. tools/collect_profiles.sh r3_end r3_06   (gpurun_out/r3_end/* -> profiles/r3_06_*)
set -e
o=gpurun_out/$1; p=profiles/$2
cp $o/bench_c2.json ${p}_bench_c2.json
cp $o/stats_c2/k_kernel_stats.csv ${p}_c2_kernel_stats.csv
cp $o/stats_c4/k_kernel_stats.csv ${p}_c4_kernel_stats.csv
cp $o/pmc_c4_sq_summary.txt ${p}_c4_sq_counters.txt
cp $o/pmc_FETCH_SIZE_summary.txt ${p}_pmc_FETCH_SIZE_summary.txt
cp $o/pmc_WRITE_SIZE_summary.txt ${p}_pmc_WRITE_SIZE_summary.txt
grep -v amdgpu.ids $o/kbench_gemm.txt > ${p}_kbench_gemm.txt
grep -v amdgpu.ids $o/kbench_attn_ln.txt > ${p}_kbench_attn_ln.txt
grep -v amdgpu.ids $o/forward_only.txt > ${p}_forward_only.txt
cp $o/ab_sched.txt ${p}_gemm_items_same_binary_ab.txt
grep -v amdgpu.ids $o/train_cli.txt > ${p}_train_cli.txt
{ echo "## tools/decode_bench.py"; grep -v amdgpu.ids $o/decode_bench.txt; echo; echo "## COMPOSER_DECODE_V1=1 tools/decode_bench.py"; grep -v amdgpu.ids $o/decode_bench_v1.txt
  echo; echo "## tools/decode_diag.py"; grep -v amdgpu.ids $o/decode_diag.txt; echo; echo "## tools/decode_trace.py (eager launches under rocprofv3 --kernel-trace)"; cat $o/decode_trace.txt
  echo; echo "## tools/ubench/bin/graph_chain"; cat $o/graph_chain.txt; } > ${p}_decode.txt
grep -E "passed|failed" $o/pytest_gpu.log | tail -1 > ${p}_gpu_suite_tail.txt
python tools/make_traffic_json.py ${p}_pmc_FETCH_SIZE_summary.txt ${p}_pmc_WRITE_SIZE_summary.txt profiles/hbm_traffic.json
ls ${p}_*
