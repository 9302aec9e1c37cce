#!/usr/bin/env python3
"""Gradients of the raw-row backward pass (COMPOSER_LN_FUSED=3) against the unfused path at benchmark shapes: per parameter, the largest
difference relative to the largest unfused entry.   python tools/ln_raw_grad_probe.py [c2b32|c2|c4] [dropout]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from composer_amd import _lib
from composer_amd.transformer import Transformer
name = sys.argv[1] if len(sys.argv) > 1 else "c2b32"
p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
E, H, L, T, B = {"c2": (512, 8, 6, 1024, 128), "c2b32": (512, 8, 6, 1024, 32), "c4": (768, 12, 12, 2048, 32), "small": (512, 8, 2, 256, 96)}[name]
rng = np.random.default_rng(0)
x = rng.integers(0, 390, (B, T), dtype=np.int32); y = rng.integers(0, 390, (B, T), dtype=np.int32)
res = {}
for mode in ("0", "3", "2"):
    os.environ["COMPOSER_LN_FUSED"] = mode
    m = Transformer(390, E, T, L, H, attention_dropout_rate=p, residual_dropout_rate=p, dtype="bf16", seed=7, max_batch=B, max_seq=T)
    m.initialize_parameters(0)
    loss, acc = m.loss_and_grads(x, y)
    res[mode] = (loss, {n: m.get_parameter(n, _lib.KIND_GRAD).astype(np.float64) for n in m.parameter_names})
    m.close()
print(name, "dropout", p, "losses", {k: v[0] for k, v in res.items()})
ref = res["0"][1]
for mode in ("3", "2"):
    worst = sorted(((np.abs(res[mode][1][n] - ref[n]).max() / (np.abs(ref[n]).max() + 1e-30), n) for n in ref), reverse=True)
    print("mode", mode, "worst:", ["%s %.3g" % (n, w) for w, n in worst[:6]])
