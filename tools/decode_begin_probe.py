#!/usr/bin/env python3
"""What `generate` spends outside the per-token chain: time of generate(prompt, 1) (cmp_decode_begin: weight transposes, prefill,
graph capture + instantiate, first id) against generate(prompt, 1024), BASELINE config 5's model."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from composer_amd.transformer import Transformer
V, E, H, L, W = 390, 512, 8, 6, 2048
m = Transformer(V, E, W, L, H, attention_dropout_rate=0.0, residual_dropout_rate=0.0, dtype="fp32", seed=0, max_batch=1, max_seq=64)
prompt = np.random.default_rng(0).integers(0, V, 10)
m.generate(prompt, 8, temperature=1.0, mode="kv", seed=1)
for n in (1, 1, 2, 1024, 1024):
    t0 = time.perf_counter(); m.generate(prompt, n, temperature=1.0, mode="kv", seed=1); dt = time.perf_counter() - t0
    print("generate(%4d): %8.3f ms" % (n, dt * 1e3))
