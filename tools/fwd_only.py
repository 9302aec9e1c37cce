#!/usr/bin/env python3
"""N inference forward passes (cmp_eval_step) of C2 or C4 for a profiler run:  python3 tools/fwd_only.py [c2|c4] [passes]
(`rocprofv3 --kernel-trace --stats -- python3 tools/fwd_only.py c2 10`; COMPOSER_LN_FUSED=0 selects the unfused block path)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from composer_amd.transformer import Transformer
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
E, H, L, T, B = {"c2": (512, 8, 6, 1024, 128), "c4": (768, 12, 12, 2048, 32)}[name]
m = Transformer(390, E, T, L, H, attention_dropout_rate=0.0, residual_dropout_rate=0.0, dtype="bf16", seed=0, max_batch=B, max_seq=T)
rng = np.random.default_rng(0)
ds = [(rng.integers(0, 390, (B, T), dtype=np.int32), rng.integers(0, 390, (B, T), dtype=np.int32))]
for _ in range(2): m.evaluate(ds)
t0 = time.perf_counter()
for _ in range(n): m.evaluate(ds)
print("%s forward: %.3f ms per pass" % (name, 1e3 * (time.perf_counter() - t0) / n))
m.close()
