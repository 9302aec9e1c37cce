#!/usr/bin/env python3
"""Host-side AddressSanitizer run of the library's argument-checking / error paths (CPU container only: GPU ASan and xnack
builds are not available on the pool).  Build + run:
    python -m composer_amd.build --asan
    LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang --print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0 python tools/asan_probe.py
Every call below must come back with a negative status and a message, and ASan must stay silent."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["COMPOSER_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "composer_amd", "lib", "libcomposer_hip_asan.so")
from composer_amd import _lib
lib = _lib.load()
print("version", lib.cmp_version(), "devices", lib.cmp_device_count())
h = C.c_void_p()
rc = lib.cmp_ctx_create(0, C.byref(h))
print("ctx_create rc", rc, _lib.last_error()[:80])
rc = lib.cmp_param_count(None, None)
print("param_count(null) rc", rc, _lib.last_error()[:60])
rc = lib.cmp_k_gemm(None, 1, 0, 0, 4, 4, 0, None, 8, None, 8, None, 8, None, 0, None, 0, None, 0, 0, 1, 0.0, 0, 0, 0)
print("gemm K=0 rc", rc, _lib.last_error()[:60])
buf = (C.c_float * 4)()
print("dp_allreduce_test(null ctx) rc", lib.cmp_dp_allreduce_test(None, buf, 4))
print("decode_steps(null) rc", lib.cmp_decode_steps(None, 1, None))
print("asan probe done")
