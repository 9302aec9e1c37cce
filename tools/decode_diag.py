#!/usr/bin/env python3
"""Per-kernel-class cost of the decode chain: time per token with one class left out of the captured graph
(COMPOSER_DECODE_DIAG_SKIP; the ids are garbage in those runs) subtracted from the full chain's."""
import os, sys, time, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "one":
    import numpy as np
    from composer_amd.transformer import Transformer
    V, E, H, L, W = 390, 512, 8, 6, 2048
    m = Transformer(V, E, W, L, H, attention_dropout_rate=0.0, residual_dropout_rate=0.0, dtype="fp32", seed=0, max_batch=1, max_seq=64)
    prompt = np.random.default_rng(0).integers(0, V, 10)
    m.generate(prompt, 64, temperature=1.0, mode="kv", seed=1)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); m.generate(prompt, 1024, temperature=1.0, mode="kv", seed=1); best = min(best, time.perf_counter() - t0)
    print(best / 1024 * 1e6)
    sys.exit(0)
names = ["LN1+c_attn", "attention", "combine+c_proj", "LN2+c_fc+gelu", "mlp c_proj", "LN_f+logits", "sampler"]
def run(skip):
    env = dict(os.environ, COMPOSER_DECODE_DIAG_SKIP=str(skip))
    return float(subprocess.run([sys.executable, __file__, "one"], env=env, capture_output=True, text=True).stdout.strip().split("\n")[-1])
full = run(0)
print("full chain: %.1f us/token" % full)
for i, n in enumerate(names):
    t = run(1 << i)
    cnt = 6 if i < 5 else 1
    print("%-16s %.2f us per launch (%d per token, %.1f us of the token)" % (n, (full - t) / cnt, cnt, full - t))
