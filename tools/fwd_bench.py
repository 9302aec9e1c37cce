#!/usr/bin/env python3
"""Forward-only rate (north_star target: fraction of the bf16 MFMA peak on the attention + FFN forward).
Times cmp_eval_step (forward + loss, dropout off; the ids upload and the per-call sync are inside, <1 %) for
  C2: 6L/8H/d512, seq 1024      C4: 12L/12H/d768, seq 2048
and reports forward model FLOPs (L*(24E^2 + 2ET) + 2EV per token, causal attention on the unmasked half) per second."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from composer_amd.transformer import Transformer

V = 390
for name, E, H, L, T, B in (("C2 6L/8H/d512 seq1024", 512, 8, 6, 1024, 128), ("C4 12L/12H/d768 seq2048", 768, 12, 12, 2048, 32)):
    m = Transformer(V, E, T, L, H, attention_dropout_rate=0.0, residual_dropout_rate=0.0, dtype="bf16", seed=0, max_batch=B, max_seq=T)
    rng = np.random.default_rng(0)
    x = rng.integers(0, V, (B, T), dtype=np.int32); y = rng.integers(0, V, (B, T), dtype=np.int32)
    ds = [(x, y)]
    for _ in range(3): m.evaluate(ds)
    ts = []
    for _ in range(30):                      # per-call times, median: one host hiccup in a 10-call mean moved it by 15 %
        t0 = time.perf_counter()
        m.evaluate(ds)
        ts.append(time.perf_counter() - t0)
    dt = float(np.median(ts))
    fl_tok = L * (24 * E * E + 2 * E * T) + 2 * E * V
    tf = B * T * fl_tok / dt / 1e12
    print("%-26s B=%3d: forward %.2f ms, %.2f M tok/s, %.0f TFLOP/s = %.1f %% of 2.5 PF bf16 dense" % (name, B, dt * 1e3, B * T / dt / 1e6, tf, 100 * tf / 2500))
    m.close()
