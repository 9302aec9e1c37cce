#!/usr/bin/env python3
"""One-off fuzzing of the attention kernels (forward, dQ, dK/dV, fused bias sums) through the kernel-level parity check of
tests/test_gpu_kernels.py on random shapes:  python tests/extra/fuzz_attention.py [N] [seed]   (COMPOSER_ATTN64 selects the opt-in forwards)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_kernels as K


def main():
    from composer_amd import _lib
    lib = _lib.load(); _lib.require_gpu()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for _ in range(n):
        dtype = int(rng.choice([K.FP32, K.BF16]))
        D = int(rng.choice([16, 32, 64, 128]))
        H = int(rng.integers(1, 5)); B = int(rng.integers(1, 4))
        T = int(rng.integers(1, 900 if rng.random() < 0.3 else 200))
        p = float(rng.choice([0.0, 0.1, 0.4]))
        try:
            K.test_attention_fwd_bwd(lib, dtype, B, T, H, D, p)
        except Exception as ex:
            bad += 1
            print("FAIL", dict(dtype=dtype, B=B, T=T, H=H, D=D, p=p), type(ex).__name__, str(ex)[:200], flush=True)
    print("%d cases, %d failures" % (n, bad))
    sys.exit(min(bad, 100))


if __name__ == "__main__":
    main()
