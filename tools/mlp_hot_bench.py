#!/usr/bin/env python3
"""The MLP c_proj forward GEMM (K = 4E) with its A operand (g) hot in the Infinity Cache against cold, at a token count whose g fits the
cache (32 768 tokens: 134 MB) -- what a launch-fused MLP pair (g consumed out of L2 / cache, never streamed from HBM) could gain per item."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from composer_amd import _lib
lib = _lib.load(); _lib.require_gpu()
E = int(sys.argv[1]) if len(sys.argv) > 1 else 512
M = 32768
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.Generator().manual_seed(0)
A = torch.randn(M, 4 * E, generator=g).to(torch.bfloat16).cuda()
W = (0.05 * torch.randn(E, 4 * E, generator=g)).to(torch.bfloat16).cuda()
b = torch.randn(E, generator=g).cuda()
r = torch.randn(M, E, generator=g).to(torch.bfloat16).cuda()
out = torch.zeros(M, E, dtype=torch.bfloat16, device="cuda")
flush = torch.zeros(1 << 28, dtype=torch.float32, device="cuda")
def run(): assert lib.cmp_k_gemm(st(), 1, 0, 1, M, E, 4 * E, P(A), 4 * E, P(W), 4 * E, P(out), E, P(b), 0, None, 0, P(r), E, 0, 1, 0.0, 0, 0, 8) == 0
for name, pre in (("hot (re-read)", lambda: None), ("cold (1 GiB flush)", lambda: flush.add_(1.0)), ("hot (re-read)", lambda: None)):
    ts = []
    for i in range(12):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        if i >= 2: ts.append(e0.elapsed_time(e1) * 1e3)
    print("E=%d MLP c_proj [%d x %d x %d], A %s: %.1f us (min %.1f)" % (E, M, E, 4 * E, name, float(np.median(ts)), min(ts)))
