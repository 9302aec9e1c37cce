#!/usr/bin/env python3
"""Which torch thread count / batch gives the fastest CPU restatement on this host (input for bench.py's cpu_baseline)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import transformer_oracle as O
from oracle.torch_restatement import TorchTrainer
V, E, H, L, T = 390, 512, 8, 6, 1024
cfg = O.Config(V, E, T, L, H)
params = O.init_params(V, E, T, L, seed=0, dtype=np.float32)
rng = np.random.default_rng(0)
print("cpu_count", os.cpu_count(), "default threads", torch.get_num_threads())
for th in (8, 16, 32, 64, 128):
    if th > (os.cpu_count() or 1): break
    torch.set_num_threads(th)
    for B in (1, 4):
        tt = TorchTrainer(cfg, params)
        x, y = O.synthetic_batch(rng, V, B, T)
        tt.train_step(x, y, 1e-3)
        t0 = time.time(); n = 0
        while time.time() - t0 < 4: tt.train_step(x, y, 1e-3); n += 1
        print("threads %3d B=%d: %.0f tok/s" % (th, B, B * T * n / (time.time() - t0)), flush=True)
