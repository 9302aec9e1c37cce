#!/usr/bin/env python3
"""One-off fuzzing of the attention launch plans (attention.hip attn_job / attn_plan_u): shapes with B*H a multiple of 8 and enough
query blocks that the paired grid, the single-block tail and the plain grid are all taken; every (batch, head, 128-position
block) must be visited and a few groups are checked in full against float64 (tests/test_gpu_kernels.py check_attention_groups).
    python tests/extra/fuzz_attention_plans.py [N] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_kernels as K


def main():
    from composer_amd import _lib
    lib = _lib.load(); _lib.require_gpu()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for _ in range(n):
        dtype = int(rng.choice([K.FP32, K.BF16, K.BF16]))
        D = int(rng.choice([16, 32, 64, 64, 128]))
        H = int(rng.choice([1, 2, 4, 8, 12, 16]))
        rows = 8 * int(rng.integers(1, 13))                    # B*H
        if rows % H:
            H = 8 if rows % 8 == 0 else 1
        B = rows // H
        T = int(rng.integers(129, 2049 if D <= 64 else 1025))
        if B * T * H * D > 48 * 1024 * 1024:
            T = max(129, 48 * 1024 * 1024 // (B * H * D))
        p = float(rng.choice([0.0, 0.1]))
        groups = sorted({(0, 0), (B - 1, H - 1), (int(rng.integers(0, B)), int(rng.integers(0, H))), ((rows // 2) // H, (rows // 2) % H)})
        try:
            K.check_attention_groups(lib, B, H, D, T, dtype, p, groups)
        except Exception as ex:
            bad += 1
            print("FAIL", dict(dtype=dtype, B=B, H=H, D=D, T=T, p=p), type(ex).__name__, str(ex)[:200], flush=True)
    print("%d cases, %d failures" % (n, bad))
    sys.exit(min(bad, 100))


if __name__ == "__main__":
    main()
