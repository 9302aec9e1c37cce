#!/usr/bin/env python3
"""Does the ROW PITCH of the streamed operand matter to the persistent 256x256 GEMM?  Activations are [tokens, 512] bf16: 1 KiB
rows, a power of two -- if the L2 channel interleave maps a 256-row x 128-byte stage onto a few channels, a padded pitch would
spread it.  A[M, K] with lda = K + pad, W^T[N, K] with ldb = K + padb; forward layout (tb = 1)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from composer_amd import _lib
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)

def run(M, N, K, pad, padb, padc=0):
    A = (torch.randn(M, K + pad, device="cuda")).to(torch.bfloat16)
    B = (torch.randn(N, K + padb, device="cuda")).to(torch.bfloat16)
    Cm = torch.zeros(M, N + padc, device="cuda", dtype=torch.bfloat16)
    def f():
        rc = lib.cmp_k_gemm(st(), 1, 0, 1, M, N, K, P(A), K + pad, P(B), K + padb, P(Cm), N + padc, None, 0, None, 0, None, 0, 0, 1, 0.0, 0, 0, 0)
        assert rc == 0, lib.cmp_last_error()
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): f()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / 20
    print("M=%6d N=%5d K=%5d  lda=K+%-4d ldb=K+%-4d ldc=N+%-4d %8.1f us  %7.1f TFLOP/s" % (M, N, K, pad, padb, padc, us, 2.0 * M * N * K / us / 1e6))

torch.zeros(1, device="cuda")
M = 131072
run(M, 2048, 512, 0, 0, 0); run(M, 2048, 512, 0, 0, 0)          # warm-up (clock, allocator)
for (N, K) in ((1536, 512),):
    for pad, padb, padc in ((0, 0, 0), (0, 0, 64), (0, 0, 0), (0, 0, 64), (64, 64, 64), (0, 0, 0), (0, 0, 128), (0, 0, 32), (0, 0, 0), (0, 0, 64)):
        run(M, N, K, pad, padb, padc)
