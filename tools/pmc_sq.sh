#!/bin/bash
# SQ counter passes over a kernel micro-benchmark for one or more library builds (A/B on one box):
#   bash tools/pmc_sq.sh <tag> <kbench target: attn|gemm|gemmfwd|ln> <kernel-name filter> <lib> [<lib> ...]
#   (lib = file name under composer_amd/lib without .so)
tag=$1; what=$2; flt=$3; shift 3
export TMPDIR=/tmp
PA="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
PB="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAVES"
for l in "$@"; do
  for p in A B; do
    if [ $p = A ]; then C="$PA"; else C="$PB"; fi
    out=gpurun_out/$tag/pmc_${l}_$p
    mkdir -p $out
    KB_B=128 KB_ITERS=3 COMPOSER_HIP_LIB=composer_amd/lib/$l.so timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $out -o k -- python3 tools/kbench.py $what > $out.log 2>&1
    python3 tools/pmc_summary.py $out $flt > gpurun_out/$tag/pmc_${l}_${p}_summary.txt
    find $out -name "*.db" -delete; rm -rf $out
  done
done
