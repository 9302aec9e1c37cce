#!/usr/bin/env python3
"""Device and host memory before / after creating, using and closing models in a loop (GPU box):  python tools/leak_probe.py [rounds]"""
import os, sys, resource
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from composer_amd.transformer import Transformer


def synthetic_batch(rng, V, B, T):
    seq = rng.integers(0, V, size=(B, T + 1), dtype=np.int32)
    return np.ascontiguousarray(seq[:, :-1]), np.ascontiguousarray(seq[:, 1:])

WHAT = os.environ.get("LEAK_WHAT", "train,async,eval,fwd,gen,dp").split(",")      # which uses to include (to localise a leak)

def one(i):
    if "oom" in WHAT:                       # a model whose parameter buffers cannot be allocated: refused, nothing left behind
        try:
            Transformer(390, 2048, 12_000_000, 1, 16, dtype="fp32", seed=i, max_batch=1, max_seq=8)
            raise SystemExit("the 390 GB model was created?")
        except Exception as e:
            assert "out of memory" in str(e).lower() or "hipMalloc" in str(e), str(e)[:200]
    if "wsoom" in WHAT:                     # parameters fit, the workspace for max_batch * max_seq tokens does not
        big = Transformer(390, 512, 1024, 6, 8, dtype="bf16", seed=i, max_batch=8192, max_seq=1024)
        big.initialize_parameters(0)
        xb = np.zeros((1, 16), np.int32)
        for _ in range(2):
            try:
                big.train_step(xb, xb, 1e-3)
                raise SystemExit("an 8192 x 1024-token workspace was allocated?")
            except Exception as e:
                assert "out of memory" in str(e).lower(), str(e)[:200]
        big.close()
    V, E, H, L, W, B = 390, 128, 4, 2, 64, 4
    m = Transformer(V, E, W, L, H, dtype="bf16" if i % 2 else "fp32", seed=i, max_batch=B, max_seq=W)
    m.initialize_parameters(i)
    x, y = synthetic_batch(np.random.default_rng(i), V, B, W)
    if "train" in WHAT: m.train_step(x, y, 1e-3)
    if "async" in WHAT:
        t = m.train_step_async(x, y, 1e-3); m.step_metrics(t)
    if "eval" in WHAT: m.evaluate([(x, y)])
    if "fwd" in WHAT:
        lg, pres = m(x[:, :10]); _ = pres[0]
    if "fwdonly" in WHAT:
        lg, pres = m(x[:, :10])
    if "fwdT" in WHAT:
        lg, pres = m(x)
    if "gen" in WHAT:
        m.generate(x[0, :5], 8, temperature=1.0, mode="kv")
        m.generate(x[0, :5], 4, temperature=0.0, mode="literal")
    if "dp" in WHAT and i % 3 == 0:
        m.init_data_parallel(0, 1, Transformer.new_unique_id()); m.train_step(x, y, 1e-3)
    m.close()

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    torch.zeros(1, device="cuda")
    for i in range(5): one(i)                       # warm-up: library state, RCCL, allocator pools
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]; rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    for i in range(n): one(100 + i)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]; rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    print("device free: %.1f MiB -> %.1f MiB (%.2f MiB per model); host max RSS: %.1f MiB -> %.1f MiB over %d models"
          % (free0 / 2**20, free1 / 2**20, (free0 - free1) / 2**20 / n, rss0 / 1024, rss1 / 1024, n))

if __name__ == "__main__":
    main()
