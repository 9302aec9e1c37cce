#!/usr/bin/env python3
"""The c_attn / c_fc forward GEMMs of C2 (or C4 with `c4`) with and without the LayerNorm fold epilogue, ON THE SAME OPERANDS
(same residency of A): what the fold costs inside the kernel, apart from where its A operand comes from.
    python tools/fold_bench.py [c2|c4]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from composer_amd import _lib
lib = _lib.load(); _lib.require_gpu()
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
E, M = {"c2": (512, 131072), "c4": (768, 65536)}[cfg]
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.Generator().manual_seed(0)
x = torch.randn(M, E, generator=g).to(torch.bfloat16).cuda()
NPs = E // 256
seg = x.float().reshape(M, NPs, 256)
mu = seg.mean(-1)
part = torch.stack([mu, ((seg - mu[..., None]) ** 2).sum(-1)], -1).contiguous()
for name, N, act in (("c_attn", 3 * E, 0), ("c_fc", 4 * E, 1)):
    W = (0.05 * torch.randn(N, E, generator=g)).to(torch.bfloat16).cuda()
    b = torch.randn(N, generator=g).cuda(); cs = torch.randn(N, generator=g).cuda()
    out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    aux = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda") if act else None
    def run(fold, n):
        for _ in range(n):
            if fold: lib.cmp_gemm_ln_next(P(part), NPs, 1e-5, P(cs), None, None, None)
            rc = lib.cmp_k_gemm(st(), 1, 0, 1, M, N, E, P(x), E, P(W), E, P(out), N, P(b), act, P(aux), N if act else 0, None, 0, 0, 1, 0.0, 0, 0, 8)
            assert rc == 0, lib.cmp_last_error()
    res = {}
    for fold in (0, 1, 0, 1):
        run(fold, 3); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(fold, 20); torch.cuda.synchronize()
        res.setdefault(fold, []).append(1e6 * (time.perf_counter() - t0) / 20)
    print("%s %s M=%d N=%d K=%d: plain %s us   fold %s us" % (cfg, name, M, N, E, " ".join("%.1f" % v for v in res[0]), " ".join("%.1f" % v for v in res[1])))
