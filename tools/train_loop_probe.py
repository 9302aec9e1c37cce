#!/usr/bin/env python3
"""Where a step of Transformer.train goes on the host: the dataset's own batch preparation, train_step_async (checks,
staging copy, enqueue of the step's launches) and the wait for the previous step's metrics, at the bench.py workload."""
import os, sys, time, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from composer_amd import dataset as D
from composer_amd.transformer import Transformer

B, T, V = int(os.environ.get("TB_B", "128")), 1024, 390
tmp = tempfile.mkdtemp(prefix="loop_probe_")
try:
    f = os.path.join(tmp, "a.data")
    D.write_synthetic_data_file(f, 1_000_000, seed=1)
    ds = D.load_dataset([f] * 8, B, T, shuffle=True, seed=0)
    m = Transformer(V, 512, T, 6, 8, dtype="bf16", seed=0, max_batch=B, max_seq=T)
    t_ds, t_sub, t_wait, t_all = [], [], [], []
    orig_async, orig_metrics = m.train_step_async, m.step_metrics
    def a(x, y, lr=None):
        t0 = time.perf_counter(); r = orig_async(x, y, lr); t_sub.append(time.perf_counter() - t0); return r
    def w(tk):
        t0 = time.perf_counter(); r = orig_metrics(tk); t_wait.append(time.perf_counter() - t0); return r
    m.train_step_async, m.step_metrics = a, w
    class Timed:
        def __iter__(self):
            it = iter(ds)
            while True:
                t0 = time.perf_counter()
                try: b = next(it)
                except StopIteration: return
                t_ds.append(time.perf_counter() - t0); t_all.append(time.perf_counter())
                yield b
    m.train(Timed(), (B, T), os.path.join(tmp, "log"), epochs=2, learning_rate=1e-3, save_frequency_mode="global_step",
            save_frequency=10**9, show_progress_bar=False, max_steps=50)
    iv = np.diff(t_all)[10:]
    ms = lambda v: 1e3 * float(np.median(v[10:]))
    print("step interval %.2f ms (median of %d) | dataset next() %.2f ms | train_step_async %.2f ms | step_metrics wait %.2f ms"
          % (1e3 * float(np.median(iv)), len(iv), ms(t_ds), ms(t_sub), ms(t_wait)))
    print("tokens/s %.3f M" % (B * T / float(np.median(iv)) / 1e6))
    m.close()
finally:
    shutil.rmtree(tmp, ignore_errors=True)
