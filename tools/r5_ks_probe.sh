#!/bin/bash
# key-split attention kernels: parity tests, then the default configuration with and without them (same box)
export TMPDIR=/tmp
o=gpurun_out/${1:-r5_ks}; mkdir -p $o
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention" 2>&1 | tail -5 | tee $o/tests.txt
for ks in 0 -1; do
  if [ $ks = -1 ]; then unset COMPOSER_ATTN_KS; else export COMPOSER_ATTN_KS=$ks; fi
  echo "== COMPOSER_ATTN_KS=${COMPOSER_ATTN_KS:-auto}" | tee -a $o/summary.txt
  python3 tools/default_config_probe.py 2>&1 | grep -v amdgpu | tee -a $o/summary.txt
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/ks$ks -o k -- python3 tools/default_config_steps.py > $o/ks$ks.log 2>&1
  python3 tools/kstats.py $o/ks$ks 40 | grep -i "attn\|total\|layernorm" | tee -a $o/summary.txt
done
find gpurun_out -name "*.db" -delete; find gpurun_out -name "k_kernel_trace.csv" -delete
