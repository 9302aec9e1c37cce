#!/usr/bin/env python3
"""What the fused c_attn bias-gradient column sums cost inside the attention backward kernels: every workgroup of a head adds
into the same 64 floats (4096 wave-level atomics per head and launch).  dq / dkv times with and without the bias pointer."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from composer_amd import _lib
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
B = int(os.environ.get("KB_B", "128")); H, D, T = 8, 64, 1024
E = H * D; M = B * T
torch.zeros(1, device="cuda")
qkv = torch.randn(M, 3 * E, device="cuda").to(torch.bfloat16)
o = torch.randn(M, E, device="cuda").to(torch.bfloat16)
do = torch.randn(M, E, device="cuda").to(torch.bfloat16)
dqkv = torch.zeros(M, 3 * E, device="cuda", dtype=torch.bfloat16)
lse = torch.zeros(B * H * T, device="cuda"); delta = torch.zeros(B * H * T, device="cuda")
bias = torch.zeros(3 * E, device="cuda")
for _ in range(5):
    lib.cmp_k_attn_bwd(st(), P(qkv), P(o), P(do), P(lse), P(delta), P(dqkv), B, T, H, D, 1, 1, 0.1, 1, 2)
for rep in range(3):
    for with_bias in (0, 1, 0, 1):
        res = []
        for cls in (4, 5):
            lib.cmp_prof_begin(cls)
            for _ in range(10):
                if with_bias: lib.cmp_attn_bwd_bias_next(P(bias))
                lib.cmp_k_attn_bwd(st(), P(qkv), P(o), P(do), P(lse), P(delta), P(dqkv), B, T, H, D, 1, 1, 0.1, 1, 2)
            ms, n, w = C.c_double(), C.c_int64(), C.c_double()
            lib.cmp_prof_end(C.byref(ms), C.byref(n), C.byref(w))
            res.append(1e3 * ms.value / n.value)
        print("B=%d bias=%d   dq %7.1f us   dkv %7.1f us" % (B, with_bias, res[0], res[1]))
