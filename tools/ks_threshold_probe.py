#!/usr/bin/env python3
"""Where the key-split attention kernels stop paying: the default model at batch 1..8 and a 6L/8H/d512-like small case, train steps back to
back with COMPOSER_ATTN_KS=0 / 1 (forced) -- run once per setting:  COMPOSER_ATTN_KS=1 python tools/ks_threshold_probe.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from composer_amd.transformer import Transformer
V = 390
out = []
for (E, H, L, T, B) in ((256, 16, 8, 1024, 1), (256, 16, 8, 1024, 2), (256, 16, 8, 1024, 3), (256, 16, 8, 1024, 4), (256, 16, 8, 1024, 6), (256, 16, 8, 512, 4),
                        (256, 8, 4, 1024, 2), (256, 8, 4, 2048, 1)):
    m = Transformer(V, E, T, L, H, dtype="bf16", seed=0, max_batch=B, max_seq=T)
    rng = np.random.default_rng(0)
    x = rng.integers(0, V, (B, T), dtype=np.int32); y = rng.integers(0, V, (B, T), dtype=np.int32)
    xd = torch.from_numpy(x).cuda(); yd = torch.from_numpy(y).cuda()
    for _ in range(5): m.train_step_device(xd.data_ptr(), yd.data_ptr(), B, T, 1e-3)
    m.synchronize(); t0 = time.perf_counter(); n = 60
    for i in range(n): m.train_step_device(xd.data_ptr(), yd.data_ptr(), B, T, 1e-3)
    m.synchronize(); dd = (time.perf_counter() - t0) / n
    out.append("H=%d D=%d T=%d B=%d (%d rows x %d blocks): %.3f ms" % (H, E // H, T, B, B * H, (T + 127) // 128, dd * 1e3))
    m.close()
print("COMPOSER_ATTN_KS=%s\n  " % os.environ.get("COMPOSER_ATTN_KS", "auto") + "\n  ".join(out))
