#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/${1:-r5_ksd}; mkdir -p $o
for v in base ksd1 ksd2 ksd3 ksd4 ksd7; do
  if [ $v = base ]; then unset COMPOSER_HIP_LIB; else export COMPOSER_HIP_LIB=composer_amd/lib/$v.so; fi
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $o/$v -o k -- python3 tools/ks_slope_probe.py 1024 0.1 > $o/$v.log 2>&1
  echo "== $v" | tee -a $o/summary.txt
  python3 tools/kstats.py $o/$v 40 | grep -i "attn" | cut -c1-110 | tee -a $o/summary.txt
done
find gpurun_out -name "*.db" -delete; find gpurun_out -name "k_kernel_trace.csv" -delete
