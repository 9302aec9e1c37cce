#!/usr/bin/env python3
"""Block plan of the attention forward / dQ launches (attention.hip: attn_job / attn_plan_u): time against the number of
(batch, head) rows per XCD that run as single blocks behind the paired ones.  Needs the experiments build (the override
COMPOSER_ATTN_PLAN_U is compiled only there):
    python tools/ab_build.py planu attention.hip -DCOMPOSER_EXPERIMENTS
    COMPOSER_HIP_LIB=composer_amd/lib/planu.so python tools/ubench/attn_plan_probe.py
-1 = the 2-D paired grid.  The baseline is repeated between the variants (the first launches of a process read slow).
Outputs of every variant are compared bit for bit with the baseline's (same per-block arithmetic, another launch order)."""
import ctypes as C
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from composer_amd import _lib

lib = _lib.load()
BF16 = 1
E, H, D, T = 512, 8, 64, 1024


def P(t):
    return C.c_void_p(t.data_ptr())


def st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def run(B, us_list, p=0.1):
    M = B * T
    qkv = (torch.randn(M, 3 * E, device="cuda")).to(torch.bfloat16)
    do = (torch.randn(M, E, device="cuda")).to(torch.bfloat16)
    o = torch.zeros(M, E, device="cuda", dtype=torch.bfloat16)
    dqkv = torch.zeros(M, 3 * E, device="cuda", dtype=torch.bfloat16)
    lse = torch.zeros(B * H * T, device="cuda")
    delta = torch.zeros(B * H * T, device="cuda")
    ref = None
    for _ in range(400):            # the first launches of a process read slow, and the clock settles over the first second
        lib.cmp_k_attn_bwd(st(), P(qkv), P(o), P(do), P(lse), P(delta), P(dqkv), B, T, H, D, 1, BF16, p, 1, 2)
    torch.cuda.synchronize()
    print("B=%d (B*H=%d, %d rows per XCD), dropout %.1f" % (B, B * H, B * H // 8, p))
    for u in us_list:
        os.environ["COMPOSER_ATTN_PLAN_U"] = str(u)
        fwd = lambda: lib.cmp_k_attn_fwd(st(), P(qkv), P(o), P(lse), B, T, H, D, 1, BF16, p, 1, 2)
        bwd = lambda: lib.cmp_k_attn_bwd(st(), P(qkv), P(o), P(do), P(lse), P(delta), P(dqkv), B, T, H, D, 1, BF16, p, 1, 2)
        tf = min(timeit(fwd) for _ in range(2))
        res = {}
        for cls in (4, 5):
            bwd()
            lib.cmp_prof_begin(cls)
            for _ in range(12):
                bwd()
            ms, n, w = C.c_double(), C.c_int64(), C.c_double()
            lib.cmp_prof_end(C.byref(ms), C.byref(n), C.byref(w))
            res[cls] = 1e3 * ms.value / n.value
        torch.cuda.synchronize()
        cur = (o.clone(), lse.clone(), dqkv.clone())
        if ref is None:
            ref = cur
        same = all(torch.equal(a, b) for a, b in zip(ref, cur))
        print("  U=%4d   fwd %7.1f us   dq %7.1f us   dkv %7.1f us   outputs %s" % (u, tf, res[4], res[5], "identical" if same else "DIFFER"))


if __name__ == "__main__":
    run(32, [-1, 8, 20, 28] * 3 + [-1])
    run(64, [-1, 16, 24, 40] * 2 + [-1])
    run(128, [-1, 8, 24, 40] * 2 + [-1])
    run(16, [-1, 4, 8, 12, 16] * 2 + [-1])
