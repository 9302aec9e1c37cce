#!/usr/bin/env python3
"""One-off fuzzing of cmp_k_gemm against float64 (GPU box only):  python tests/extra/fuzz_gemm.py [N] [seed]
Random dtype, operand layouts, M / N / K (tiny to a few thousand, ragged), leading-dimension padding, kernel-selection flags,
epilogue (bias, gelu with the pre-activation output, gelu' of an auxiliary input, residual, dropout) or split-K accumulation.
The reference is built from the SAME rounded operands; dropout masks come from the oracle's restatement of the counter hash."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import transformer_oracle as O
import test_gpu_kernels as K




def main():
    from composer_amd import _lib
    lib = _lib.load(); _lib.require_gpu()
    N_CASES = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    torch.zeros(1, device="cuda")
    bad = 0
    for it in range(N_CASES):
        dtype = int(rng.choice([K.FP32, K.BF16]))
        ta, tb = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        big = rng.random() < 0.3
        huge = bool(os.environ.get("FUZZ_BIG")) and rng.random() < 0.5     # FUZZ_BIG=1: shapes that reach the persistent kernels
        M = int(rng.integers(1, 3000 if big else 300)) if not huge else int(rng.integers(3000, 40000))
        Nn = int(rng.integers(1, 200 if not big else 1200)) * (8 if dtype == K.BF16 or rng.random() < 0.5 else 1)
        Kk = int(rng.integers(1, 2100 if big else 400))
        flags = int(rng.choice([0, 0, 0, 2, 4, 8, 16, 48]))
        mode = rng.choice(["plain", "bias", "gelu", "gelugrad", "resid", "drop", "splitk"])
        if mode == "splitk" and not big:
            Kk = int(rng.integers(64, 4000))
        # operands with padded leading dimensions (multiples of 8 for bf16)
        def pad(n):
            p = n + int(rng.integers(0, 3)) * 8
            return (p + 7) // 8 * 8 if dtype == K.BF16 else p
        a = rng.standard_normal((Kk, M) if ta else (M, Kk)).astype(np.float32)
        b = rng.standard_normal((Nn, Kk) if tb else (Kk, Nn)).astype(np.float32)
        def mk(x):
            ld = pad(x.shape[1])
            t = torch.zeros(x.shape[0], ld)
            t[:, :x.shape[1]] = torch.from_numpy(x)
            return K.dev(t, dtype)
        A, Bm = mk(a), mk(b)
        A64 = A.double()[:, :a.shape[1]]; B64 = Bm.double()[:, :b.shape[1]]
        ref = ((A64.t() if ta else A64) @ (B64.t() if tb else B64)).cpu().numpy()      # float64 on the GPU (vendor library): reference only
        kw = dict(flags=flags)
        tol = K.TOL[dtype]
        try:
            if mode == "splitk":
                sk = int(rng.integers(2, 20))
                C0 = torch.zeros(M, Nn, device="cuda", dtype=torch.float32)
                out = K.gemm(lib, dtype, ta, tb, A, Bm, M, Nn, Kk, out_fp32=True, splitk=sk, C0=C0, **kw)
            else:
                bias = K.dev(torch.from_numpy(rng.standard_normal(Nn).astype(np.float32)), K.FP32) if mode != "plain" and rng.random() < 0.8 else None
                if bias is not None:
                    ref = ref + bias.double().cpu().numpy()[None]
                aux = None; resid = None; act = 0; p = 0.0; seed = 0; rs = 0
                if mode == "gelu":
                    act = 1; aux = torch.zeros(M, Nn, device="cuda", dtype=K.tdt(dtype))
                    pre = ref.copy(); ref = O.gelu(ref)
                elif mode == "gelugrad":
                    act = 2; auxh = rng.standard_normal((M, Nn)).astype(np.float32)
                    aux = K.dev(torch.from_numpy(auxh), dtype)
                    ref = ref * O.gelu_grad(aux.double().cpu().numpy())
                elif mode in ("resid", "drop"):
                    if mode == "drop":
                        p = float(rng.choice([0.1, 0.5])); seed = int(rng.integers(0, 1 << 30)); rs = int(rng.integers(0, 1 << 20))
                        keep = O.dropout_keep_rows(seed, rs, M, Nn, p).reshape(M, Nn)
                        ref = ref * keep / (1.0 - p)
                    rh = rng.standard_normal((M, Nn)).astype(np.float32)
                    resid = K.dev(torch.from_numpy(rh), dtype)
                    ref = ref + resid.double().cpu().numpy()
                if dtype == K.BF16 and Nn % 8:
                    continue
                out = K.gemm(lib, dtype, ta, tb, A, Bm, M, Nn, Kk, bias=bias, act=act, aux=aux, resid=resid, p_drop=p, seed=seed, rng=rs, **kw)
                if mode == "gelu":
                    e = K.rel_err(aux, torch.from_numpy(pre))
                    assert e < tol, ("aux", e)
            e = K.rel_err(out, torch.from_numpy(ref))
            assert e < tol * (3 if mode in ("gelugrad",) else 1), ("out", e)
        except Exception as ex:
            bad += 1
            print("FAIL", dict(dtype=dtype, ta=ta, tb=tb, M=M, N=Nn, K=Kk, flags=flags, mode=str(mode), lda=A.shape[1], ldb=Bm.shape[1]), "->",
                  type(ex).__name__, str(ex)[:200], flush=True)
    print("%d cases, %d failures" % (N_CASES, bad))
    sys.exit(min(bad, 100))


if __name__ == "__main__":
    main()
