#!/bin/bash
# where a D = 16 attention forward tile goes at the default configuration: measurement builds of attention.hip (ATTN_DIAG ladders)
export TMPDIR=/tmp
o=gpurun_out/${1:-r5_a16}; mkdir -p $o
for v in base attn_diag1 attn_diag4 attn_diag5; do
  if [ $v = base ]; then unset COMPOSER_HIP_LIB; else export COMPOSER_HIP_LIB=composer_amd/lib/$v.so; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/$v -o k -- python3 tools/default_config_steps.py > $o/$v.log 2>&1
  echo "== $v" | tee -a $o/summary.txt
  python3 tools/kstats.py $o/$v 40 | grep -i "attn\|total" | tee -a $o/summary.txt
done
find gpurun_out -name "*.db" -delete; find gpurun_out -name "k_kernel_trace.csv" -delete
