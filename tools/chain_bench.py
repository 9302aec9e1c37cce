#!/usr/bin/env python3
"""How fast the c_fc forward GEMM of C2 runs depending on WHO wrote its A operand just before it (HIP events around the consumer only):
    python tools/chain_bench.py
writers: the LayerNorm kernel, a residual-epilogue GEMM with non-temporal stores, the same with default-policy stores (the
statistics-emitting kind), a torch copy kernel, nobody (A re-read back to back), a 1 GiB flush."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from composer_amd import _lib
lib = _lib.load(); _lib.require_gpu()
# CHAIN_PRODUCER_LIB: the WRITER GEMMs come from another build of the library (store / load policy experiments); the consumer stays
plib = lib
if os.environ.get("CHAIN_PRODUCER_LIB"):
    plib = C.CDLL(os.path.abspath(os.environ["CHAIN_PRODUCER_LIB"]))
    for nm in ("cmp_k_gemm", "cmp_gemm_ln_next"):
        getattr(plib, nm).restype, getattr(plib, nm).argtypes = _lib.SIGNATURES[nm]
E, M, N = 512, 131072, 2048
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.Generator().manual_seed(0)
src = torch.randn(M, E, generator=g).to(torch.bfloat16).cuda()
att = torch.randn(M, E, generator=g).to(torch.bfloat16).cuda()
A = torch.zeros(M, E, dtype=torch.bfloat16, device="cuda")
Wp = (0.05 * torch.randn(E, E, generator=g)).to(torch.bfloat16).cuda()
W = (0.05 * torch.randn(N, E, generator=g)).to(torch.bfloat16).cuda()
b = torch.randn(N, generator=g).cuda(); bp = torch.randn(E, generator=g).cuda()
gam = torch.ones(E).cuda(); bet = torch.zeros(E).cuda()
mean = torch.zeros(M).cuda(); rstd = torch.zeros(M).cuda()
part = torch.zeros(M, 2, 2).cuda()
out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
flush = torch.zeros(1 << 28, dtype=torch.float32, device="cuda")


def w_ln(): assert lib.cmp_k_layernorm_fwd(st(), P(src), P(gam), P(bet), P(A), P(mean), P(rstd), M, E, 1e-5, 1) == 0
def w_gemm_nt(): assert plib.cmp_k_gemm(st(), 1, 0, 1, M, E, E, P(att), E, P(Wp), E, P(A), E, P(bp), 0, None, 0, P(src), E, 0, 1, 0.0, 0, 0, 8) == 0
def w_gemm_plain():
    plib.cmp_gemm_ln_next(None, 0, 0.0, None, None, None, P(part))
    assert plib.cmp_k_gemm(st(), 1, 0, 1, M, E, E, P(att), E, P(Wp), E, P(A), E, P(bp), 0, None, 0, P(src), E, 0, 1, 0.0, 0, 0, 8) == 0, lib.cmp_last_error()
def w_copy(): A.copy_(src)
def w_none(): pass
def w_flush(): flush.add_(1.0)
def consumer(): assert lib.cmp_k_gemm(st(), 1, 0, 1, M, N, E, P(A), E, P(W), E, P(out), N, P(b), 1, None, 0, None, 0, 0, 1, 0.0, 0, 0, 8) == 0


for name, w in (("LayerNorm kernel", w_ln), ("GEMM, non-temporal stores", w_gemm_nt), ("GEMM, default-policy stores", w_gemm_plain),
                ("torch copy", w_copy), ("nobody (re-read)", w_none), ("1 GiB flush", w_flush), ("LayerNorm kernel", w_ln),
                ("GEMM, default-policy stores", w_gemm_plain)):
    ts = []
    for i in range(12):
        w()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); consumer(); e1.record(); torch.cuda.synchronize()
        if i >= 2: ts.append(e0.elapsed_time(e1) * 1e3)
    # the writer's own time, alone
    wt = []
    for i in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); w(); e1.record(); torch.cuda.synchronize()
        if i >= 2: wt.append(e0.elapsed_time(e1) * 1e3)
    print("A written by %-28s: c_fc %.1f us (min %.1f)   writer itself %.1f us" % (name, float(np.median(ts)), min(ts), float(np.median(wt))))
