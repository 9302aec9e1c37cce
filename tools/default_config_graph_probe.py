#!/usr/bin/env python3
"""What a hipGraph of the WHOLE train step would buy: the step is stream-captured, instantiated and replayed back to
back (cmp_train_step_graph_probe; frozen dropout masks / Adam iteration: timing only) against the same steps as stream launches.
    python tools/default_config_graph_probe.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from composer_amd.transformer import Transformer
from composer_amd import _lib
lib = _lib.load()
V = 390
for (E, H, L, T, B) in ((256, 16, 8, 1024, 1), (256, 16, 8, 1024, 8), (512, 8, 6, 1024, 4)):
    m = Transformer(V, E, T, L, H, dtype="bf16", seed=0, max_batch=B, max_seq=T)
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.integers(0, V, (B, T), dtype=np.int32)).cuda(); y = torch.from_numpy(rng.integers(0, V, (B, T), dtype=np.int32)).cuda()
    for _ in range(10): m.train_step_device(x.data_ptr(), y.data_ptr(), B, T, 1e-3)
    m.synchronize(); n = 100; t0 = time.perf_counter()
    for _ in range(n): m.train_step_device(x.data_ptr(), y.data_ptr(), B, T, 1e-3)
    m.synchronize(); stream_ms = 1e3 * (time.perf_counter() - t0) / n
    nk, no, ms = C.c_int(0), C.c_int(0), C.c_float(0)
    _lib.check(lib.cmp_train_step_graph_probe(m._h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), B, T, C.byref(nk), C.byref(no), n, C.byref(ms)))
    print("E=%d L=%d T=%d B=%d: stream launches %.3f ms/step; graph replay %.3f ms/step (%d kernel nodes, %d others)" % (E, L, T, B, stream_ms, ms.value, nk.value, no.value))
    m.close()
