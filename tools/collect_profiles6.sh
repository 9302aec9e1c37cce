#!/bin/bash
# Copies what is judged from a tools/profile_round6.sh run into profiles/:  tools/collect_profiles5.sh <tag> <prefix>
set -e
o=gpurun_out/$1; p=profiles/$2
for c in c2 c2b32 c4; do cp $o/stats_$c/k_kernel_stats.csv ${p}_${c}_kernel_stats.csv; done
cp $o/hbm_traffic.json profiles/hbm_traffic.json
cp $o/hbm_traffic.json ${p}_hbm_traffic.json
cp $o/pmc_c4_sq_summary.txt ${p}_c4_sq_counters.txt
grep -v amdgpu.ids $o/kbench_gemm.txt > ${p}_kbench_gemm.txt
grep -v amdgpu.ids $o/kbench_attn_ln.txt > ${p}_kbench_attn_ln.txt
grep -v amdgpu.ids $o/decode_bench.txt > ${p}_decode.txt
for f in bench_c2.json pytest_gpu_tail.txt forward_only.txt fwd_kernels_c2.txt fwd_kernels_c4.txt train_cli.txt default_config.txt; do cp $o/$f ${p}_$f; done
ls ${p}_*
for f in train_kernels_c2.txt ab_ln_raw.txt bench_dp1.json allreduce_only.json; do grep -v amdgpu.ids $o/$f > ${p}_$f; done
