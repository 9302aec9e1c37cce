#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/${1:-r5_ring2}; mkdir -p $o
for w in 0 2 5 8 0 2 5 8; do
  export COMPOSER_GEMM_RING=$w
  echo "== COMPOSER_GEMM_RING=$w" | tee -a $o/summary.txt
  python3 tools/default_config_probe.py 2>&1 | grep -v amdgpu | tee -a $o/summary.txt
done
