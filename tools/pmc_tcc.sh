#!/bin/bash
# HBM-side traffic and L2 hit rate of a kernel micro-benchmark (separate passes: FETCH_SIZE and WRITE_SIZE do not fit one):
#   bash tools/pmc_tcc.sh <tag> <kbench target> <kernel-name filter> [lib]
tag=$1; what=$2; flt=$3; l=${4:-libcomposer_hip}
export TMPDIR=/tmp
mkdir -p gpurun_out/$tag
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum"; do
  n=$(echo $C | tr ' ' '_')
  out=gpurun_out/$tag/pmc_$n
  mkdir -p $out
  KB_B=${KB_B:-128} KB_ITERS=3 COMPOSER_HIP_LIB=composer_amd/lib/$l.so timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $out -o k -- python3 tools/kbench.py $what > $out.log 2>&1
  python3 tools/pmc_summary.py $out $flt > gpurun_out/$tag/pmc_${n}_summary.txt
  find $out -name "*.db" -delete; rm -rf $out
done
cat gpurun_out/$tag/pmc_*_summary.txt
