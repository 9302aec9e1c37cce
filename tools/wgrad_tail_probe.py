#!/usr/bin/env python3
"""Grouped weight-gradient launch (gemm.hip: gemm_wgrad_group_la_kernel) against the token count: at few tokens the launch IS its fixed
part (partial-tile stores, tickets, the last arrivers' sums), at the benchmark depth the k-loops dominate.  One line per (shape set, K).
    python tools/wgrad_tail_probe.py                      # the form the library defaults to
    COMPOSER_WGRAD_TAIL=atomic python tools/wgrad_tail_probe.py      # round 4's float-atomic form, same binary
"""
import ctypes as C
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from composer_amd import _lib

lib = _lib.load()


def run(name, E, Ks):
    shapes = [(4 * E, E), (E, 4 * E), (E, E), (E, 3 * E)]
    n = len(shapes)
    for K in Ks:
        As = [(torch.randn(K, m, device="cuda") * 0.1).to(torch.bfloat16) for m, _ in shapes]
        Bs = [(torch.randn(K, k, device="cuda") * 0.1).to(torch.bfloat16) for _, k in shapes]
        Cs = [torch.zeros(m, k, device="cuda") for m, k in shapes]
        vp, ip = C.c_void_p * n, C.c_int * n
        args = (vp(*[a.data_ptr() for a in As]), ip(*[m for m, _ in shapes]), vp(*[b.data_ptr() for b in Bs]), ip(*[k for _, k in shapes]),
                vp(*[c.data_ptr() for c in Cs]), ip(*[k for _, k in shapes]), ip(*[m for m, _ in shapes]), ip(*[k for _, k in shapes]))
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

        def one():
            rc = lib.cmp_k_wgrad_group(st, n, *args, K)
            assert rc == 0, lib.cmp_last_error()
        for _ in range(3):
            one()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                one()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 100.0)
        us = sorted(ts)[len(ts) // 2]
        fl = sum(2.0 * m * k * K for m, k in shapes)
        print("%s K=%6d  %8.1f us  %7.1f TFLOP/s   (%s)" % (name, K, us, fl / us / 1e6, os.environ.get("COMPOSER_WGRAD_TAIL", "last-arriver")), flush=True)


if __name__ == "__main__":
    run("C2 block (E=512, 48 tiles) ", 512, [1024, 8192, 32768, 131072])
    run("C4 block (E=768, 108 tiles)", 768, [1024, 8192, 65536])
