#!/bin/bash
# HBM / L2 counters of the forward attention kernel under COMPOSER_ATTN64=$1: tools/attn_mem_pmc.sh <mode> <outdir>
mode=$1; out=$2; mkdir -p $out
export COMPOSER_ATTN64=$mode KB_B=128
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$n -o k -- python3 tools/kbench.py attnfwd > $out/$n.log 2>&1
  python3 tools/pmc_summary.py $out/$n attn_fwd >> $out/mem_summary_$mode.txt 2>&1
  rm -f $out/$n/k_kernel_trace.csv $out/$n/k_counter_collection.csv
done
