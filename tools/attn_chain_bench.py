#!/usr/bin/env python3
"""The attention c_proj forward GEMM of C2 timed alone behind different writers of its A operand `att`: the attention forward kernel
(which also streams the 403 MB qkv tensor through the cache), nobody (re-read), a 1 GiB flush."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from composer_amd import _lib
lib = _lib.load(); _lib.require_gpu()
B, T, H, D, E = 128, 1024, 8, 64, 512
M = B * T
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.Generator().manual_seed(0)
qkv = (0.5 * torch.randn(M, 3 * E, generator=g)).to(torch.bfloat16).cuda()
att = torch.zeros(M, E, dtype=torch.bfloat16, device="cuda")
lse = torch.zeros(B * H * T, device="cuda")
W = (0.05 * torch.randn(E, E, generator=g)).to(torch.bfloat16).cuda()
b = torch.randn(E, generator=g).cuda()
r = torch.randn(M, E, generator=g).to(torch.bfloat16).cuda()
out = torch.zeros(M, E, dtype=torch.bfloat16, device="cuda")
flush = torch.zeros(1 << 28, dtype=torch.float32, device="cuda")
def w_attn(): assert lib.cmp_k_attn_fwd(st(), P(qkv), P(att), P(lse), B, T, H, D, 1, 1, 0.1, 1, 2) == 0
def consumer(): assert lib.cmp_k_gemm(st(), 1, 0, 1, M, E, E, P(att), E, P(W), E, P(out), E, P(b), 0, None, 0, P(r), E, 0, 1, 0.1, 3, 4, 8) == 0
for name, w in (("attention forward", w_attn), ("nobody (re-read)", lambda: None), ("1 GiB flush", lambda: flush.add_(1.0)), ("attention forward", w_attn)):
    ts = []
    for i in range(12):
        w()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); consumer(); e1.record(); torch.cuda.synchronize()
        if i >= 2: ts.append(e0.elapsed_time(e1) * 1e3)
    print("att written by %-20s: c_proj %.1f us (min %.1f)" % (name, float(np.median(ts)), min(ts)))
