#!/usr/bin/env python3
"""In-kernel timeline of the deep-pipeline GEMM (s_memtime stamps of workgroup 17, wave 0): where an item's time goes."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.kbench as kb
import torch
kb._lib.require_gpu(); torch.zeros(1, device="cuda")
lib = kb.lib
B = int(os.environ.get("KB_B", "128"))
M, N, K = B * 1024, int(os.environ.get("N", "1536")), int(os.environ.get("K", "512"))
KIND = os.environ.get("KIND", "plain")          # plain | gelu | resid | wgrad (C[M,N] += A[K,M]^T B[K,N], split-K atomics)
A = kb.rnd(M, K); Bm = kb.rnd(K, N); Cm = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16); bias = torch.randn(N, device="cuda")
aux = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16) if KIND == "gelu" else None
res = kb.rnd(M, N) if KIND == "resid" else None
WG = KIND == "wgrad"
if WG:
    Mw, Nw, Kw = int(os.environ.get("M", "512")), N, B * 1024
    A = kb.rnd(Kw, Mw); Bm = kb.rnd(Kw, Nw); Cm = torch.zeros(Mw, Nw, device="cuda")
    SPL = int(os.environ.get("SPLITK", "16"))
def run():
    if WG:
        assert lib.cmp_k_gemm(kb.st(), 1, 1, 0, Mw, Nw, Kw, kb.P(A), Mw, kb.P(Bm), Nw, kb.P(Cm), Nw, None, 0, None, 0, None, 0, 1, SPL, 0.0, 0, 0, 16) == 0
        return
    assert lib.cmp_k_gemm(kb.st(), 1, 0, 0, M, N, K, kb.P(A), K, kb.P(Bm), N, kb.P(Cm), N, kb.P(bias), 1 if KIND == "gelu" else 0,
                          kb.P(aux) if aux is not None else None, N if aux is not None else 0,
                          kb.P(res) if res is not None else None, N if res is not None else 0, 0, 1,
                          0.1 if KIND == "resid" else 0.0, 5, 3, 16) == 0
for _ in range(3): run()
buf = torch.zeros(512, device="cuda", dtype=torch.int64)
lib.cmp_gemm_set_stamps(C.c_void_p(buf.data_ptr()))
run(); torch.cuda.synchronize()
lib.cmp_gemm_set_stamps(None)
v = buf.cpu().numpy().reshape(-1, 2)
v = v[v[:, 0] != 0]
names = {1: "item start", 2: "before first wait", 3: "after vmcnt wait (stage0)", 4: "after barrier", 10: "sync: arrive", 11: "sync: after vmcnt", 12: "sync: after barrier",
         13: "sync: DMA issued", 20: "loop done", 21: "after final barrier", 22: "next item's stages issued", 30: "epilogue issued", 31: "store drain done"}
t0 = v[0, 1]
prev = t0
for i, (k, t) in enumerate(v[:150]):
    if 10 <= k <= 13 and not os.environ.get("TL_ALL"): prev = t; continue
    print("%-28s  +%7d  (delta %6d cycles)" % (names.get(int(k), str(k)), t - t0, t - prev))
    prev = t
