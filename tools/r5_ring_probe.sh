#!/bin/bash
# four-stage ring in the 128x128 GEMM kernel: parity, then the default configuration with and without it (same box)
export TMPDIR=/tmp
o=gpurun_out/${1:-r5_ring}; mkdir -p $o
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm" 2>&1 | tail -5 | tee $o/tests.txt
for w in 0 1; do
  export COMPOSER_GEMM_RING=$w
  echo "== COMPOSER_GEMM_RING=$w" | tee -a $o/summary.txt
  python3 tools/default_config_probe.py 2>&1 | grep -v amdgpu | tee -a $o/summary.txt
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/w$w -o k -- python3 tools/default_config_steps.py > $o/w$w.log 2>&1
  python3 tools/kstats.py $o/w$w 40 | grep -i "gemm\|total" | tee -a $o/summary.txt
  python3 tools/small_gemm_probe.py 2>&1 | grep -v amdgpu | tee -a $o/summary.txt
done
find gpurun_out -name "*.db" -delete; find gpurun_out -name "k_kernel_trace.csv" -delete
