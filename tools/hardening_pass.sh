#!/bin/bash
# Every fuzzer and operational probe once, on the GPU box (about three minutes):  bash tools/hardening_pass.sh [seed]
# One line per tool; anything but "0 failures" / "0 mismatches" / "0.00 MiB per model" / "equal" needs a look.
s=${1:-1}
f() { "$@" 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -1; }
FUZZ_STD=0.02 f timeout 600 python tests/extra/fuzz_shapes.py 200 $s
FUZZ_WIDE=1 FUZZ_STD=0.02 f timeout 900 python tests/extra/fuzz_shapes.py 60 $((s+1))
f timeout 600 python tests/extra/fuzz_stateful.py 40 40 $((s+2))
FUZZ_DP=1 f timeout 600 python tests/extra/fuzz_stateful.py 20 40 $((s+3))
FUZZ_DTYPE=bf16 f timeout 600 python tests/extra/fuzz_stateful.py 20 40 $((s+4))
f timeout 600 python tests/extra/fuzz_gemm.py 200 $((s+5))
f timeout 600 python tests/extra/fuzz_attention.py 100 $((s+6))
f timeout 900 python tests/extra/fuzz_attention_plans.py 40 $((s+7))
f timeout 300 python tools/leak_probe.py 40
f timeout 300 python tools/soak_probe.py 2000
COMPOSER_DETERMINISTIC=1 THREAD_MODELS=3 f timeout 300 python tools/thread_probe.py 60
COMPOSER_DETERMINISTIC=1 f timeout 300 python tools/shared_ctx_probe.py
