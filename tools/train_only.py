#!/usr/bin/env python3
"""N train steps of C2 / C2 at B=32 / C4 for a profiler run:  python3 tools/train_only.py [c2|c2b32|c4] [steps]
(`rocprofv3 --kernel-trace --stats -- python3 tools/train_only.py c2 12`; COMPOSER_LN_FUSED=2 sends training passes down the
LayerNorm-fused block path, =0 keeps every pass off it)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from composer_amd.transformer import Transformer
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
E, H, L, T, B = {"c2": (512, 8, 6, 1024, 128), "c2b32": (512, 8, 6, 1024, 32), "c4": (768, 12, 12, 2048, 32)}[name]
m = Transformer(390, E, T, L, H, attention_dropout_rate=0.1, residual_dropout_rate=0.1, dtype="bf16", seed=1000, max_batch=B, max_seq=T)
m.initialize_parameters(0)
rng = np.random.default_rng(1234)
seq = rng.integers(0, 390, size=(2, B, T + 1), dtype=np.int32)
xs = [torch.from_numpy(np.ascontiguousarray(seq[i, :, :-1])).cuda() for i in range(2)]
ys = [torch.from_numpy(np.ascontiguousarray(seq[i, :, 1:])).cuda() for i in range(2)]
for i in range(3): m.train_step_device(xs[i % 2].data_ptr(), ys[i % 2].data_ptr(), B, T, 1e-3)
m.synchronize()
t0 = time.perf_counter()
for i in range(n): m.train_step_device(xs[i % 2].data_ptr(), ys[i % 2].data_ptr(), B, T, 1e-3)
m.synchronize()
print("%s train: %.3f ms per step (%d steps), loss %.4f" % (name, 1e3 * (time.perf_counter() - t0) / n, n, m.last_metrics()[0]))
m.close()
