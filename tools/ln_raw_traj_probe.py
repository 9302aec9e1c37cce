#!/usr/bin/env python3
"""Loss trajectories from the seeded initialisation, LayerNorm-fused training forms against the unfused path, each form twice (the
float atomics make two runs of one form differ too): python tools/ln_raw_traj_probe.py [cfg] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from composer_amd.transformer import Transformer
name = sys.argv[1] if len(sys.argv) > 1 else "c2b32"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 18
E, H, L, T, B = {"c2": (512, 8, 6, 1024, 128), "c2b32": (512, 8, 6, 1024, 32), "c4": (768, 12, 12, 2048, 32)}[name]
rng = np.random.default_rng(1234)
seq = rng.integers(0, 390, size=(2, B, T + 1), dtype=np.int32)
xs, ys = [seq[i, :, :-1].copy() for i in range(2)], [seq[i, :, 1:].copy() for i in range(2)]
for mode in ("0", "0", "3", "3", "2"):
    os.environ["COMPOSER_LN_FUSED"] = mode
    m = Transformer(390, E, T, L, H, attention_dropout_rate=0.1, residual_dropout_rate=0.1, dtype="bf16", seed=1000, max_batch=B, max_seq=T)
    m.initialize_parameters(0)
    traj = [m.train_step(xs[i % 2], ys[i % 2], 1e-3)[0] for i in range(n)]
    m.close()
    print("mode", mode, " ".join("%.3f" % t for t in traj))
