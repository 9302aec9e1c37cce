#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/${1:-r5_bp}; mkdir -p $o
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention" 2>&1 | tail -2 | tee $o/tests.txt
for bp in 0 100000; do
  export COMPOSER_ATTN_BIAS_PASS=$bp
  echo "== COMPOSER_ATTN_BIAS_PASS=$bp" | tee -a $o/summary.txt
  python3 tools/default_config_probe.py 2>&1 | grep -v amdgpu | tee -a $o/summary.txt
  python3 tools/ks_threshold_probe.py 2>&1 | grep -v amdgpu | tee -a $o/summary.txt
  python3 bench.py --no-cpu-baseline --no-decode --no-extras --batch 32 --steps 20 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 B=32: %.3f ms/step' % d['ms_per_step'])" | tee -a $o/summary.txt
  python3 bench.py --no-cpu-baseline --no-decode --no-extras --batch 8 --steps 20 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 B=8: %.3f ms/step' % d['ms_per_step'])" | tee -a $o/summary.txt
done
unset COMPOSER_ATTN_BIAS_PASS
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/def -o k -- python3 tools/default_config_steps.py > $o/def.log 2>&1
python3 tools/kstats.py $o/def 14 | cut -c1-120 | tee -a $o/summary.txt
find gpurun_out -name "*.db" -delete; find gpurun_out -name "k_kernel_trace.csv" -delete
