#!/usr/bin/env python3
"""Two models driven from two Python threads at once (ctypes releases the GIL inside every call, so the native code runs
concurrently: each model has its own context and streams, the library's process-wide state is shared): the per-step losses
and greedy ids of each thread must equal the same sequence run alone.   python tools/thread_probe.py [steps]"""
import os, sys, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from composer_amd.transformer import Transformer


def synthetic_batch(rng, V, B, T):
    seq = rng.integers(0, V, size=(B, T + 1), dtype=np.int32)
    return np.ascontiguousarray(seq[:, :-1]), np.ascontiguousarray(seq[:, 1:])


def work(tag, steps, out):
    try:
        work_(tag, steps, out)
    except Exception as e:
        import traceback
        out[tag + "_error"] = traceback.format_exc()


def work_(tag, steps, out):
    cfgs = {"a": (390, 128, 4, 2, 64, 4, "bf16"), "b": (1000, 96, 2, 3, 40, 3, "fp32")}
    V, E, H, L, W, B, dt = cfgs[tag]
    res = []
    for gen in range(int(os.environ.get("THREAD_MODELS", "1"))):       # successive models: each one's first generate captures its graph
        m = Transformer(V, E, W, L, H, dtype=dt, seed=5 + gen, max_batch=B, max_seq=W, attention_dropout_rate=0.1, residual_dropout_rate=0.1)
        m.initialize_parameters(3 + gen)
        rng = np.random.default_rng(7 + gen)
        for i in range(steps):
            x, y = synthetic_batch(rng, V, B, W)
            res.append(m.train_step(x, y, 1e-3)[0])
            if i % 5 == 0:
                res.append(float(m.evaluate([(x, y)])[0]))
                res.extend(m.generate(x[0, :6], 5, temperature=0.0, mode="kv").tolist())
                t = m.train_step_async(x, y, 1e-3); res.append(m.step_metrics(t)[0])
            if i % 7 == 0:                                             # forward, presents (temporary device buffers), past, weight I/O
                lg, pres = m(x[:, :W - 1]); p0 = np.array(pres[0])
                l2, _ = m(x, past=pres)
                res.append(float(np.abs(lg).sum())); res.append(float(np.abs(p0).sum())); res.append(float(np.abs(l2).sum()))
                n0 = m.parameter_names[3]; m.set_parameter(n0, m.get_parameter(n0))
        m.close()
    out[tag] = res


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    alone = {}
    work("a", steps, alone); work("b", steps, alone)
    bad = 0
    for rep in range(3):
        both = {}
        ts = [threading.Thread(target=work, args=(t, steps, both)) for t in ("a", "b")]
        for t in ts: t.start()
        for t in ts: t.join()
        for tag in ("a", "b"):
            if tag + "_error" in both:
                print("EXCEPTION in thread", tag, both[tag + "_error"][-600:])
            a, b = np.array(alone[tag], float), np.array(both.get(tag, []), float)
            # COMPOSER_DETERMINISTIC=1 (no float atomics): bitwise; otherwise the summation order of the atomics differs from run to
            # run and long trajectories drift apart on their own -- use few steps
            det = os.environ.get("COMPOSER_DETERMINISTIC") == "1"
            ok = a.shape == b.shape and (np.array_equal(a, b) if det else np.allclose(a, b, rtol=2e-3 if tag == "a" else 2e-5, atol=0))
            if not ok:
                bad += 1
                d = np.abs(a - b).max() if a.shape == b.shape else -1
                print("MISMATCH rep %d model %s: max abs difference %s" % (rep, tag, d))
    print("3 concurrent runs of 2 models x %d steps: %d mismatches" % (steps, bad))
    sys.exit(bad)


if __name__ == "__main__":
    main()
