#!/usr/bin/env python3
"""What paces the 128x128-tile GEMMs of the reference's default configuration (M = 1 024 tokens, 16-64 tiles on 256 CUs):
time against K, tile count and split-K (fp32 atomic output), back-to-back launches timed with HIP events.
    python tools/small_gemm_probe.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from composer_amd import _lib
lib = _lib.load(); _lib.require_gpu()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.Generator().manual_seed(0)


def run(M, N, K, splitk=1, out_fp32=0, reps=50, flags=0):
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    W = (0.05 * torch.randn(N, K, generator=g)).to(torch.bfloat16).cuda()
    out = torch.zeros(M, N, dtype=torch.float32 if out_fp32 else torch.bfloat16, device="cuda")
    def go():
        rc = lib.cmp_k_gemm(st(), 1, 0, 1, M, N, K, P(A), K, P(W), K, P(out), N, None, 0, None, 0, None, 0, out_fp32, splitk, 0.0, 0, 0, flags)
        assert rc == 0, lib.cmp_last_error()
    for _ in range(5): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): go()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for M, N in ((1024, 256), (1024, 1024), (2048, 512), (8192, 256)):
    print("M=%d N=%d (%d tiles):" % (M, N, (M // 128) * (N // 128)), "  ".join("K=%d %.1f us" % (K, run(M, N, K)) for K in (64, 256, 512, 1024, 2048, 4096)))
print("split-K, fp32 atomic output, M=1024 N=256 K=1024:", "  ".join("x%d %.1f us" % (sk, run(1024, 256, 1024, sk, 1)) for sk in (1, 2, 4, 8)))
print("split-K, fp32 atomic output, M=1024 N=256 K=4096:", "  ".join("x%d %.1f us" % (sk, run(1024, 256, 4096, sk, 1)) for sk in (1, 2, 4, 8, 16)))
print("register-staged 128x128 kernel (flags=2: global_load -> ds_write, two stages), M=1024 N=256:", "  ".join("K=%d %.1f us" % (K, run(1024, 256, K, flags=2)) for K in (256, 1024, 2048, 4096)))
