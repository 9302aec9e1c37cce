#!/usr/bin/env python3
"""Same-box A/B of whole train steps between builds of the library (a kernel change is accepted on THIS, not on kbench):
    python tools/ab_step.py composer_amd/lib/r3_baseline.so composer_amd/lib/libcomposer_hip.so [--rounds 3] [--cfg c2,c2b32,c4]
Each arm runs in its own process (the library path is fixed at import), arms alternate A B A B ..., per configuration the median
ms/step of every arm is printed.  AB_ENV_<i>="K=V,K=V" adds environment variables to arm i."""
import json, os, subprocess, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = {"c2": (512, 8, 6, 1024, 128), "c2b32": (512, 8, 6, 1024, 32), "c4": (768, 12, 12, 2048, 32)}

def child(cfgs, steps):
    sys.path.insert(0, ROOT)
    import time, torch
    from composer_amd.transformer import Transformer
    out = {}
    for name in cfgs:
        E, H, L, T, B = CFG[name]
        m = Transformer(390, E, T, L, H, attention_dropout_rate=0.1, residual_dropout_rate=0.1, dtype="bf16", seed=1000, max_batch=B, max_seq=T)
        m.initialize_parameters(0)
        rng = np.random.default_rng(1234)
        seq = rng.integers(0, 390, size=(2, B, T + 1), dtype=np.int32)
        xs = [torch.from_numpy(np.ascontiguousarray(seq[i, :, :-1])).cuda() for i in range(2)]
        ys = [torch.from_numpy(np.ascontiguousarray(seq[i, :, 1:])).cuda() for i in range(2)]
        for i in range(3):
            m.train_step_device(xs[i % 2].data_ptr(), ys[i % 2].data_ptr(), B, T, 1e-3)
        m.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            m.train_step_device(xs[i % 2].data_ptr(), ys[i % 2].data_ptr(), B, T, 1e-3)
        m.synchronize()
        out[name] = 1e3 * (time.perf_counter() - t0) / steps
        out[name + "_loss"] = m.last_metrics()[0]
        m.close()
    print("AB_RESULT " + json.dumps(out), flush=True)

def main():
    if sys.argv[1] == "--child":
        return child(sys.argv[2].split(","), int(sys.argv[3]))
    libs = [a for a in sys.argv[1:] if a.endswith(".so")]
    rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 3
    cfgs = sys.argv[sys.argv.index("--cfg") + 1] if "--cfg" in sys.argv else "c2,c2b32,c4"
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 15
    res = {l: [] for l in range(len(libs))}
    for r in range(rounds):
        for i, l in enumerate(libs):
            env = dict(os.environ, COMPOSER_HIP_LIB=os.path.abspath(l))
            for kv in filter(None, os.environ.get("AB_ENV_%d" % i, "").split(",")):
                k, v = kv.split("=", 1); env[k] = v
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", cfgs, str(steps)], env=env, capture_output=True, text=True)
            line = [x for x in p.stdout.splitlines() if x.startswith("AB_RESULT ")]
            if not line:
                print("arm %d failed:\n%s\n%s" % (i, p.stdout[-2000:], p.stderr[-2000:])); continue
            res[i].append(json.loads(line[0][10:]))
    for c in cfgs.split(","):
        print(c + ":  " + "   ".join("%s %s ms (median %.3f) loss %.4f" % (os.path.basename(libs[i]), " ".join("%.3f" % r[c] for r in res[i]),
              float(np.median([r[c] for r in res[i]])) if res[i] else float("nan"), res[i][-1][c + "_loss"] if res[i] else float("nan")) for i in range(len(libs))))

if __name__ == "__main__":
    main()
