#!/usr/bin/env python3
"""Reference point only (not used by the product): what the vendor library behind torch.matmul (hipBLASLt / rocBLAS,
hand-scheduled assembly kernels) reaches on this model's GEMM shapes, plain bf16 GEMM without epilogue."""
import torch, time
M = 128 * 1024
E = 512
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for name, m, n, k, ta, tb in (("fwd c_attn", M, 3 * E, E, 0, 0), ("fwd attn c_proj", M, E, E, 0, 0), ("fwd c_fc", M, 4 * E, E, 0, 0), ("fwd mlp c_proj", M, E, 4 * E, 0, 0),
                               ("dgrad c_fc  (B^T)", M, E, 4 * E, 0, 1), ("dgrad mlp (B^T)", M, 4 * E, E, 0, 1),
                               ("wgrad c_fc  (A^T)", E, 4 * E, M, 1, 0), ("wgrad mlp  (A^T)", 4 * E, E, M, 1, 0)):
    a = torch.randn((k, m) if ta else (m, k), device="cuda", dtype=torch.bfloat16)
    b = torch.randn((n, k) if tb else (k, n), device="cuda", dtype=torch.bfloat16)
    A = a.t() if ta else a
    B = b.t() if tb else b
    us = t(lambda: torch.matmul(A, B))
    print("%-20s M=%6d N=%5d K=%6d  %8.1f us  %7.1f TFLOP/s" % (name, m, n, k, us, 2.0 * m * n * k / us / 1e6))
