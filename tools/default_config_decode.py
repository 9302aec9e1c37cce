#!/usr/bin/env python3
"""`composer generate` at the reference's default model (default_config.yml: E=256, L=8, H=16, window 1024): 1014 tokens from a
10-id prompt at temperature 1.0, both decode modes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from composer_amd.transformer import Transformer
V, E, H, L, W = 390, 256, 16, 8, 1024
m = Transformer(V, E, W, L, H, attention_dropout_rate=0.0, residual_dropout_rate=0.0, dtype="fp32", seed=0, max_batch=1, max_seq=64)
prompt = np.random.default_rng(0).integers(0, V, 10)
for mode in ("kv", "literal"):
    m.generate(prompt, 64, temperature=1.0, mode=mode, seed=1)
    t0 = time.perf_counter(); m.generate(prompt, 1014, temperature=1.0, mode=mode, seed=1); dt = time.perf_counter() - t0
    print("default config decode, mode %-8s %.1f us/token, %.0f tokens/s" % (mode, 1e6 * dt / 1014, 1014 / dt))
