#!/usr/bin/env python3
"""Decode throughput (BASELINE config 5): `generate` 1024 tokens at temperature 1.0 from a 10-id prompt with the KV
cache and the hipGraph-captured per-token step, C2 model (6L/8H/d512) with window 2048; also the literal mode."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from composer_amd.transformer import Transformer

V, E, H, L, W = 390, 512, 8, 6, 2048
m = Transformer(V, E, W, L, H, attention_dropout_rate=0.0, residual_dropout_rate=0.0, dtype="fp32", seed=0, max_batch=1, max_seq=64)
prompt = np.random.default_rng(0).integers(0, V, 10)
out = {}
for mode in ("kv", "literal"):
    for graph in (("1",) if os.environ.get("DECODE_EAGER_ONLY") else ("0", "1")):
        os.environ["COMPOSER_NO_GRAPH"] = graph
        m.generate(prompt, 64, temperature=1.0, mode=mode, seed=1)       # warm-up (allocations, graph instantiate)
        t0 = time.perf_counter()
        ids = m.generate(prompt, 1024, temperature=1.0, mode=mode, seed=1)
        dt = time.perf_counter() - t0
        out["%s_%s" % (mode, "eager" if graph == "1" else "graph")] = {"tokens_per_s": 1024 / dt, "us_per_token": 1e6 * dt / 1024}
print(json.dumps(out))
