#!/usr/bin/env python3
"""One model, many thousands of calls of every entry point in a loop: device memory and host RSS at the start and the end
(GPU box):  python tools/soak_probe.py [iterations]"""
import os, sys, resource
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from composer_amd.transformer import Transformer


def synthetic_batch(rng, V, B, T):
    seq = rng.integers(0, V, size=(B, T + 1), dtype=np.int32)
    return np.ascontiguousarray(seq[:, :-1]), np.ascontiguousarray(seq[:, 1:])


def rss():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * 4096 / 2**20


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    V, E, H, L, W, B = 390, 64, 2, 2, 48, 4
    m = Transformer(V, E, W, L, H, dtype="bf16", seed=1, max_batch=B, max_seq=W, attention_dropout_rate=0.1, residual_dropout_rate=0.1)
    m.initialize_parameters(0)
    rng = np.random.default_rng(0)
    def body(i):
        T = int(rng.integers(1, W + 1)); Bq = int(rng.integers(1, B + 1))
        x, y = synthetic_batch(rng, V, Bq, T)
        m.train_step(x, y, 1e-3)
        t = m.train_step_async(x, y, 1e-3); m.step_metrics(t)
        if i % 4 == 0: m.evaluate([(x, y)])
        if i % 5 == 0:
            lg, pres = m(x); p0 = pres[0]
            if T < W: m(np.concatenate([x, y[:, -1:]], 1), past=pres)
        if i % 7 == 0: m.generate(x[0, :min(T, 8)], 6, temperature=1.0, mode="kv", seed=i)
        if i % 11 == 0: m.generate(x[0, :min(T, 8)], 3, temperature=0.0, mode="literal")
        if i % 50 == 0: m.load_state_dict(m.state_dict())
    for i in range(200): body(i)
    torch.cuda.synchronize()
    f0, r0 = torch.cuda.mem_get_info()[0] / 2**20, rss()
    for i in range(n): body(i)
    torch.cuda.synchronize()
    f1, r1 = torch.cuda.mem_get_info()[0] / 2**20, rss()
    print("%d iterations: device free %.1f -> %.1f MiB, host RSS %.1f -> %.1f MiB; last loss %.4f" % (n, f0, f1, r0, r1, m.last_metrics()[0]))
    m.close()


if __name__ == "__main__":
    main()
