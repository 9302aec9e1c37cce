#!/bin/bash
# SQ counters of the forward attention kernel under COMPOSER_ATTN64=$1 (off | force | pipe): tools/attn_pmc.sh <mode> <outdir>
mode=$1; out=$2; mkdir -p $out
export COMPOSER_ATTN64=$mode KB_B=128
for c in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$n -o k -- python3 tools/kbench.py attnfwd > $out/$n.log 2>&1
  python3 tools/pmc_summary.py $out/$n attn_fwd >> $out/summary_$mode.txt 2>&1
  rm -f $out/$n/k_kernel_trace.csv $out/$n/k_counter_collection.csv
done
