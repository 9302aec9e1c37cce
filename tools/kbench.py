#!/usr/bin/env python3
"""Kernel micro-benchmarks at the BASELINE C2 shapes (M = 32*1024 tokens, E=512, H=8, D=64), through the C ABI.
    python tools/kbench.py [gemm] [attn] [ln] ...      prints one line per kernel: avg us, TFLOP/s or GB/s
Used to iterate on a kernel with a 40-second GPU round trip; bench.py remains the headline measurement."""
import ctypes as C
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from composer_amd import _lib

lib = _lib.load()
BF16 = 1
GFLAGS = 0
B = int(os.environ.get('KB_B', '32'))
M, E, H, D, T = B * 1024, 512, 8, 64, 1024


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters     # us


def rnd(*shape, dtype=torch.bfloat16, scale=1.0):
    return (torch.randn(*shape, device="cuda") * scale).to(dtype)


def gemm_case(name, ta, tb, m, n, k, splitk=1, out_fp32=False, bias=False, act=0, resid=False, flags=0):
    A = rnd(k, m) if ta else rnd(m, k)
    Bm = rnd(n, k) if tb else rnd(k, n)
    Cm = torch.zeros(m, n, device="cuda", dtype=torch.float32 if out_fp32 else torch.bfloat16)
    bi = torch.randn(n, device="cuda") if bias else None
    aux = torch.zeros(m, n, device="cuda", dtype=torch.bfloat16) if act else None
    rs = rnd(m, n) if resid else None
    def run():
        rc = lib.cmp_k_gemm(st(), BF16, ta, tb, m, n, k, P(A), A.shape[1], P(Bm), Bm.shape[1], P(Cm), n, P(bi), act, P(aux),
                            n if act else 0, P(rs), n if resid else 0, int(out_fp32), splitk, 0.0, 0, 0, flags | GFLAGS)
        assert rc == 0, lib.cmp_last_error()
    us = timeit(run)
    print("%-34s M=%6d N=%5d K=%6d splitk=%2d  %8.1f us  %7.1f TFLOP/s" % (name, m, n, k, splitk, us, 2.0 * m * n * k / us / 1e6))
    return us


def bench_gemm_fwd():
    """only the four forward GEMMs of a layer (the PMC traffic pass of tools/profile_round.sh)"""
    gemm_case("fwd c_attn (+bias)", 0, 1, M, 3 * E, E, bias=True)
    gemm_case("fwd attn c_proj (+bias,+resid)", 0, 1, M, E, E, bias=True, resid=True)
    gemm_case("fwd c_fc (+bias,gelu,aux)", 0, 1, M, 4 * E, E, bias=True, act=1)
    gemm_case("fwd mlp c_proj (+bias,+resid)", 0, 1, M, E, 4 * E, bias=True, resid=True)


def bench_gemm_diag():
    """plain [M,K]x[N,K]^T GEMMs of the 256x256 persistent kernel at three depths: per-item time = a + b * (K/64)"""
    r = []
    for m, n, k in ((131072, 512, 512), (131072, 512, 2048), (65536, 512, 8192)):
        us = gemm_case("plain", 0, 1, m, n, k)
        items_per_cu = (m // 256) * (n // 256) / 256.0
        r.append((k // 64, us / items_per_cu))
    b = (r[2][1] - r[1][1]) / (r[2][0] - r[1][0])
    print("per item: %s us;  b = %.3f us per k-step (K=64), a = %.2f us per item" % (", ".join("K=%d: %.2f" % (64 * ks, t) for ks, t in r), b, r[1][1] - b * r[1][0]))


def bench_wgrad_diag():
    """wgrad shapes (deep-pipeline kernel, split-K): contraction over M tokens"""
    for nm, m, n in (("wgrad c_fc", E, 4 * E), ("wgrad attn c_proj", E, E)):
        for sk in (16, 32):
            gemm_case(nm, 1, 0, m, n, M, splitk=sk, out_fp32=True)


def bench_gemm():
    tot = 0
    if os.environ.get("KBENCH_SLABS"):
        ws = torch.empty(64 * E * E + 64, device="cuda")
        lib.cmp_gemm_set_workspace(P(ws), ws.numel() * 4)
    # forward: the model feeds the transposed bf16 weight copy (tb = 1: both operands K-contiguous)
    tot += gemm_case("fwd c_attn (+bias)", 0, 1, M, 3 * E, E, bias=True)
    tot += gemm_case("fwd attn c_proj (+bias,+resid)", 0, 1, M, E, E, bias=True, resid=True)
    tot += gemm_case("fwd c_fc (+bias,gelu,aux)", 0, 1, M, 4 * E, E, bias=True, act=1)
    tot += gemm_case("fwd mlp c_proj (+bias,+resid)", 0, 1, M, E, 4 * E, bias=True, resid=True)
    if os.environ.get("KBENCH_BN"):      # the same with the weight as stored, [K,N] (deep-pipeline kernel)
        gemm_case("fwd c_attn, W[K,N]", 0, 0, M, 3 * E, E, bias=True)
        gemm_case("fwd attn c_proj, W[K,N]", 0, 0, M, E, E, bias=True, resid=True)
        gemm_case("fwd c_fc, W[K,N]", 0, 0, M, 4 * E, E, bias=True, act=1)
        gemm_case("fwd mlp c_proj, W[K,N]", 0, 0, M, E, 4 * E, bias=True, resid=True)
    tot += gemm_case("dgrad mlp c_proj (*gelu')", 0, 1, M, 4 * E, E, act=2)
    tot += gemm_case("dgrad c_fc", 0, 1, M, E, 4 * E)
    tot += gemm_case("dgrad attn c_proj", 0, 1, M, E, E)
    tot += gemm_case("dgrad c_attn (+resid)", 0, 1, M, E, 3 * E, resid=True)
    for nm, m, n in (("wgrad mlp c_proj", 4 * E, E), ("wgrad c_fc", E, 4 * E), ("wgrad attn c_proj", E, E), ("wgrad c_attn", E, 3 * E)):
        # model.hip::wgrad_splits
        t256 = ((m + 255) // 256) * ((n + 255) // 256)
        t128 = ((m + 127) // 128) * ((n + 127) // 128)
        s = max(1, 256 // t256) if t256 >= 8 else max(1, 768 // t128)
        s = max(2, min(s, M // 256))
        tot += gemm_case(nm, 1, 0, m, n, M, splitk=s, out_fp32=True)
    print("sum of the 12 per-layer GEMMs: %.1f us  (%.1f TFLOP/s average)" % (tot, 3 * 2.0 * M * 12 * E * E / tot / 1e6))


def bench_wgrad_group():
    """the four weight gradients of a decoder block: four split-K launches (model.hip::wgrad_splits) against ONE grouped launch"""
    shapes = [(4 * E, E), (E, 4 * E), (E, E), (E, 3 * E)]
    As = [rnd(M, m) for m, n in shapes]
    Bs = [rnd(M, n) for m, n in shapes]
    Cs = [torch.zeros(m, n, device="cuda") for m, n in shapes]
    def four():
        for (m, n), a, b, c in zip(shapes, As, Bs, Cs):
            t256 = ((m + 255) // 256) * ((n + 255) // 256)
            t128 = ((m + 127) // 128) * ((n + 127) // 128)
            sk = max(1, 256 // t256) if t256 >= 8 else max(1, 768 // t128)
            sk = max(2, min(sk, M // 256))
            rc = lib.cmp_k_gemm(st(), BF16, 1, 0, m, n, M, P(a), m, P(b), n, P(c), n, None, 0, None, 0, None, 0, 1, sk, 0.0, 0, 0, GFLAGS)
            assert rc == 0, lib.cmp_last_error()
    n = len(shapes)
    vp, ip = C.c_void_p * n, C.c_int * n
    args = (vp(*[a.data_ptr() for a in As]), ip(*[m for m, _ in shapes]), vp(*[b.data_ptr() for b in Bs]), ip(*[k for _, k in shapes]),
            vp(*[c.data_ptr() for c in Cs]), ip(*[k for _, k in shapes]), ip(*[m for m, _ in shapes]), ip(*[k for _, k in shapes]))
    def one():
        rc = lib.cmp_k_wgrad_group(st(), n, *args, M)
        assert rc == 0, lib.cmp_last_error()
    fl = sum(2.0 * m * k * M for m, k in shapes)
    for nm, fn in (("four split-K launches", four), ("one grouped launch", one), ("four split-K launches", four), ("one grouped launch", one)):
        us = timeit(fn)
        print("wgrad of a block, K=%6d: %-22s %8.1f us  %7.1f TFLOP/s" % (M, nm, us, fl / us / 1e6))


def bench_attn_fwd():
    qkv = rnd(M, 3 * E)
    o = torch.zeros(M, E, device="cuda", dtype=torch.bfloat16)
    lse = torch.zeros(B * H * T, device="cuda")
    fl = 2.0 * B * H * T * T * D
    for p in (0.0, 0.1):
        us = min(timeit(lambda: lib.cmp_k_attn_fwd(st(), P(qkv), P(o), P(lse), B, T, H, D, 1, BF16, p, 1, 2)) for _ in range(3))
        print("attn fwd  p=%.1f  %8.1f us  %7.1f TFLOP/s (causal-half flops)" % (p, us, fl / us / 1e6))


def bench_attn():
    qkv = rnd(M, 3 * E)
    o = torch.zeros(M, E, device="cuda", dtype=torch.bfloat16)
    do = rnd(M, E)
    dqkv = torch.zeros(M, 3 * E, device="cuda", dtype=torch.bfloat16)
    lse = torch.zeros(B * H * T, device="cuda")
    delta = torch.zeros(B * H * T, device="cuda")
    fl = 2.0 * B * H * T * T * D
    for p in (0.0, 0.1):
        us = timeit(lambda: lib.cmp_k_attn_fwd(st(), P(qkv), P(o), P(lse), B, T, H, D, 1, BF16, p, 1, 2))
        print("attn fwd  p=%.1f  %8.1f us  %7.1f TFLOP/s (causal-half flops)" % (p, us, fl / us / 1e6))
        us = timeit(lambda: lib.cmp_k_attn_bwd(st(), P(qkv), P(o), P(do), P(lse), P(delta), P(dqkv), B, T, H, D, 1, BF16, p, 1, 2))
        print("attn bwd  p=%.1f  %8.1f us  %7.1f TFLOP/s (2.5x fwd flops)" % (p, us, 2.5 * fl / us / 1e6))
        for cls, nm in ((4, "dq"), (5, "dkv")):
            lib.cmp_prof_begin(cls)
            for _ in range(10):
                lib.cmp_k_attn_bwd(st(), P(qkv), P(o), P(do), P(lse), P(delta), P(dqkv), B, T, H, D, 1, BF16, p, 1, 2)
            ms, n, w = C.c_double(), C.c_int64(), C.c_double()
            lib.cmp_prof_end(C.byref(ms), C.byref(n), C.byref(w))
            print("    %-4s p=%.1f  %8.1f us  %7.1f TFLOP/s" % (nm, p, 1e3 * ms.value / n.value, w.value / ms.value / 1e9))


def bench_ln():
    x, dy, rs = rnd(M, E), rnd(M, E), rnd(M, E)
    y = torch.empty_like(x); dx = torch.empty_like(x)
    g, b = torch.ones(E, device="cuda"), torch.zeros(E, device="cuda")
    mean, rstd = torch.zeros(M, device="cuda"), torch.ones(M, device="cuda")
    dg, db = torch.zeros(E, device="cuda"), torch.zeros(E, device="cuda")
    ws = torch.empty(lib.cmp_k_layernorm_bwd_ws(M, E) // 4 + 16, device="cuda")
    us = timeit(lambda: lib.cmp_k_layernorm_fwd(st(), P(x), P(g), P(b), P(y), P(mean), P(rstd), M, E, 1e-5, BF16))
    print("layernorm fwd   %8.1f us  %7.1f GB/s" % (us, M * E * 4 / us / 1e3))
    us = timeit(lambda: lib.cmp_k_layernorm_bwd(st(), P(dy), P(x), P(g), P(mean), P(rstd), P(rs), P(dx), P(dg), P(db), P(ws), M, E, BF16))
    print("layernorm bwd   %8.1f us  %7.1f GB/s (dy,x,resid in; dx out)" % (us, M * E * 8 / us / 1e3))
    big = rnd(M, 4 * E)
    out = torch.zeros(4 * E, device="cuda")
    us = timeit(lambda: lib.cmp_k_colsum(st(), P(big), 4 * E, P(out), M, 4 * E, BF16))
    print("colsum [M,4E]   %8.1f us  %7.1f GB/s" % (us, M * 4 * E * 2 / us / 1e3))


if __name__ == "__main__":
    GFLAGS = int(os.environ.get("KBENCH_GEMM_FLAGS", "0"))
    what = sys.argv[1:] or ["gemm", "attn", "ln"]
    _lib.require_gpu()
    torch.zeros(1, device="cuda")
    if "gemm" in what:
        bench_gemm()
    if "gemmfwd" in what:
        bench_gemm_fwd()
    if "gemmdiag" in what:
        bench_gemm_diag()
    if "wgradgroup" in what:
        bench_wgrad_group()
    if "wgraddiag" in what:
        bench_wgrad_diag()
    if "attn" in what:
        bench_attn()
    if "attnfwd" in what:
        bench_attn_fwd()
    if "ln" in what:
        bench_ln()
