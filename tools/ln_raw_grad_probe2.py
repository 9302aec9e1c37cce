#!/usr/bin/env python3
"""Like ln_raw_grad_probe.py, but from weights that have MOVED (N unfused train steps first: gamma / beta away from 1 / 0), and the loss
trajectories of the three forms from those weights.   python tools/ln_raw_grad_probe2.py [cfg] [warm steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from composer_amd import _lib
from composer_amd.transformer import Transformer
name = sys.argv[1] if len(sys.argv) > 1 else "c2b32"
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 12
E, H, L, T, B = {"c2": (512, 8, 6, 1024, 128), "c2b32": (512, 8, 6, 1024, 32), "c4": (768, 12, 12, 2048, 32), "small": (512, 8, 2, 256, 96)}[name]
p = 0.1
rng = np.random.default_rng(1234)
seq = rng.integers(0, 390, size=(2, B, T + 1), dtype=np.int32)
xs, ys = [seq[i, :, :-1].copy() for i in range(2)], [seq[i, :, 1:].copy() for i in range(2)]
def make(mode):
    os.environ["COMPOSER_LN_FUSED"] = mode
    return Transformer(390, E, T, L, H, attention_dropout_rate=p, residual_dropout_rate=p, dtype="bf16", seed=1000, max_batch=B, max_seq=T)
m = make("0")
m.initialize_parameters(0)
for i in range(warm): m.train_step(xs[i % 2], ys[i % 2], 1e-3)
W = m.get_weights()
opt = m.get_optimizer_state() if hasattr(m, "get_optimizer_state") else None
m.close()
print("gamma range", float(min(W[n].min() for n in W if n.endswith("gamma"))), float(max(W[n].max() for n in W if n.endswith("gamma"))))
res = {}
for mode in ("0", "3", "2"):
    m = make(mode)
    m.set_weights(W)
    loss, acc = m.loss_and_grads(xs[0], ys[0])
    g = {n: m.get_parameter(n, _lib.KIND_GRAD).astype(np.float64) for n in m.parameter_names}
    traj = [m.train_step(xs[i % 2], ys[i % 2], 1e-3)[0] for i in range(8)]
    res[mode] = (loss, g, traj)
    m.close()
for mode in res: print("mode", mode, "loss %.5f" % res[mode][0], "traj", ["%.4f" % t for t in res[mode][2]])
ref = res["0"][1]
for mode in ("3", "2"):
    worst = sorted(((np.abs(res[mode][1][n] - ref[n]).max() / (np.abs(ref[n]).max() + 1e-30), n) for n in ref), reverse=True)
    print("mode", mode, "worst:", ["%s %.3g" % (n, w) for w, n in worst[:8]])
