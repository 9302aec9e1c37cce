#!/usr/bin/env python3
"""What the c_attn bias-gradient sums cost the attention backward at a given shape: cmp_k_attn_bwd timed with the sums not armed, armed
(fused float atomics, or the column-sum pass below the 16 MiB bound) -- run again with COMPOSER_ATTN_BIAS_PASS=0 / 100000 for the other form.
    python tools/attn_bias_tail_probe.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from composer_amd import _lib
lib = _lib.load(); _lib.require_gpu()
P = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.Generator().manual_seed(0)
print("COMPOSER_ATTN_BIAS_PASS=%s" % os.environ.get("COMPOSER_ATTN_BIAS_PASS", "(16)"))
for (B, H, D, T) in ((128, 8, 64, 1024), (32, 8, 64, 1024), (8, 8, 64, 1024), (32, 12, 64, 2048), (8, 16, 16, 1024), (1, 16, 16, 1024)):
    E = H * D
    qkv = torch.randn(B * T, 3 * E, generator=g).to(torch.bfloat16).cuda()
    do = torch.randn(B * T, E, generator=g).to(torch.bfloat16).cuda()
    o = torch.zeros(B * T, E, device="cuda", dtype=torch.bfloat16); lse = torch.zeros(B * H * T, device="cuda")
    assert lib.cmp_k_attn_fwd(st(), P(qkv), P(o), P(lse), B, T, H, D, 1, 1, 0.1, 5, 3) == 0
    dqkv = torch.zeros(B * T, 3 * E, device="cuda", dtype=torch.bfloat16); delta = torch.zeros(B * H * T, device="cuda")
    bias = torch.zeros(3 * E, device="cuda")
    def run(armed, n):
        for _ in range(n):
            if armed: lib.cmp_attn_bwd_bias_next(P(bias))
            assert lib.cmp_k_attn_bwd(st(), P(qkv), P(o), P(do), P(lse), P(delta), P(dqkv), B, T, H, D, 1, 1, 0.1, 5, 3) == 0
    res = {}
    for armed in (0, 1, 0, 1):
        run(armed, 3); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(armed, 20); e1.record(); torch.cuda.synchronize()
        res.setdefault(armed, []).append(e0.elapsed_time(e1) * 1e3 / 20)
    print("B=%d H=%d D=%d T=%d (gradient %.1f MB): backward without the sums %s us, with %s us" %
          (B, H, D, T, B * T * 3 * E * 2 / 1e6, " / ".join("%.1f" % v for v in res[0]), " / ".join("%.1f" % v for v in res[1])))
