#!/usr/bin/env python3
"""The grouped weight-gradient launch at the C4 block (E = 768, 65 536 tokens): 108 output tiles on 256 CUs."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from composer_amd import _lib
lib = _lib.load()
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
E, M = 768, 65536
shapes = [(4 * E, E), (E, 4 * E), (E, E), (E, 3 * E)]
As = [torch.randn(M, m, device="cuda").to(torch.bfloat16) for m, n in shapes]
Bs = [torch.randn(M, n, device="cuda").to(torch.bfloat16) for m, n in shapes]
Cs = [torch.zeros(m, n, device="cuda") for m, n in shapes]
n = len(shapes)
vp, ip = C.c_void_p * n, C.c_int * n
args = (vp(*[a.data_ptr() for a in As]), ip(*[m for m, _ in shapes]), vp(*[b.data_ptr() for b in Bs]), ip(*[k for _, k in shapes]),
        vp(*[c.data_ptr() for c in Cs]), ip(*[k for _, k in shapes]), ip(*[m for m, _ in shapes]), ip(*[k for _, k in shapes]))
def one():
    assert lib.cmp_k_wgrad_group(st(), n, *args, M) == 0, lib.cmp_last_error()
for _ in range(5): one()
torch.cuda.synchronize()
for rep in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): one()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / 20
    print("C4 block wgrad, one grouped launch: %.1f us  %.1f TFLOP/s" % (us, sum(2.0 * m * k * M for m, k in shapes) / us / 1e6))
