#!/usr/bin/env python3
"""Two models on ONE context (same streams), used alternately, against the same models on contexts of their own
(COMPOSER_DETERMINISTIC=1: bitwise).  python tools/shared_ctx_probe.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from composer_amd.transformer import Transformer


def synthetic_batch(rng, V, B, T):
    seq = rng.integers(0, V, size=(B, T + 1), dtype=np.int32)
    return np.ascontiguousarray(seq[:, :-1]), np.ascontiguousarray(seq[:, 1:])


def run(shared):
    cfgs = [(390, 128, 4, 2, 64, 4, "bf16"), (1000, 96, 2, 3, 40, 3, "fp32")]
    ms = []
    for k, (V, E, H, L, W, B, dt) in enumerate(cfgs):
        kw = dict(ctx=ms[0]._ctx) if (shared and ms) else {}
        m = Transformer(V, E, W, L, H, dtype=dt, seed=5 + k, max_batch=B, max_seq=W, attention_dropout_rate=0.1, residual_dropout_rate=0.1, **kw)
        m.initialize_parameters(3 + k)
        ms.append(m)
    rngs = [np.random.default_rng(7), np.random.default_rng(8)]
    res = []
    for i in range(60):
        for k, m in enumerate(ms):
            V, E, H, L, W, B, dt = cfgs[k]
            x, y = synthetic_batch(rngs[k], V, B, W)
            res.append(m.train_step(x, y, 1e-3)[0])
            if i % 4 == k:
                res.extend(m.generate(x[0, :6], 5, temperature=0.0, mode="kv").tolist())
                t = m.train_step_async(x, y, 1e-3); res.append(m.step_metrics(t)[0])
                lg, pres = m(x[:, :W - 1]); l2, _ = m(x, past=pres); res.append(float(np.abs(l2).sum()))
    for m in reversed(ms):
        m.close()
    return np.array(res, float)


a = run(False); b = run(True)
det = os.environ.get("COMPOSER_DETERMINISTIC") == "1"
ok = a.shape == b.shape and (np.array_equal(a, b) if det else np.allclose(a, b, rtol=2e-3))
print("shared context vs own contexts: %s (%d values, max abs difference %.3g)" % ("equal" if ok else "MISMATCH", len(a), np.abs(a - b).max()))
sys.exit(0 if ok else 1)
