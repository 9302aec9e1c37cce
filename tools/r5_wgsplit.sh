#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/${1:-r5_wgs}; mkdir -p $o
export COMPOSER_HIP_LIB=composer_amd/lib/wgexp.so
for c in 4 1 2 3 5 6 8 4; do
  export COMPOSER_WGRAD_SPLITS=$c
  echo "== COMPOSER_WGRAD_SPLITS=$c" | tee -a $o/summary.txt
  python3 tools/default_config_probe.py 2>&1 | grep -v amdgpu | grep "T=1024 B=1\|B=8" | tee -a $o/summary.txt
done
