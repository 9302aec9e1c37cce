#!/bin/bash
# key-split attention kernels: time against the window (256 keys per staged tile) and dropout, default model width (16 heads of 16, batch 1)
export TMPDIR=/tmp
o=gpurun_out/${1:-r5_kss}; mkdir -p $o
for cfg in "256 0.1" "512 0.1" "768 0.1" "1024 0.1" "1024 0.0" "512 0.0"; do
  set -- $cfg
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $o/t$1_$2 -o k -- python3 tools/ks_slope_probe.py $1 $2 > $o/t$1_$2.log 2>&1
  echo "== T=$1 p=$2" | tee -a $o/summary.txt
  python3 tools/kstats.py $o/t$1_$2 40 | grep -i "attn" | cut -c1-110 | tee -a $o/summary.txt
done
find gpurun_out -name "*.db" -delete; find gpurun_out -name "k_kernel_trace.csv" -delete
