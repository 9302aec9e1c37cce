#!/bin/bash
# the GPU suite's kernel / model / CLI files under the forced forms of round 5's switches (each must stay green: the switches choose speed, never results)
o=gpurun_out/${1:-r5_forced}; mkdir -p $o
run() { echo "== $*" | tee -a $o/summary.txt; env "$@" python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_cli.py tests/test_gpu_round3.py -q -m gpu -p no:cacheprovider > $o/last.log 2>&1; grep -E "passed|failed|error" $o/last.log | tail -1 | tee -a $o/summary.txt; grep -E "^FAILED|^ERROR" $o/last.log | head -5 | tee -a $o/summary.txt; }
run COMPOSER_ATTN_KS=1
run COMPOSER_ATTN_KS=0 COMPOSER_GEMM_RING=0 COMPOSER_ATTN_BIAS_PASS=0
run COMPOSER_ATTN_BIAS_PASS=100000 COMPOSER_GEMM_RING=8
