"""Generates tests/golden/transformer_*.npz from the float64 oracle (oracle/transformer_oracle.py).

Run from the repo root:  python tests/golden/make_golden.py
The oracle is a restatement (TensorFlow is not installable offline -> "parity unpinned", see the
oracle header); these files freeze its outputs so that the GPU box, which has neither the reference
nor this container's numpy RNG guarantees, checks the HIP path against fixed numbers.
All parameters are float32-representable so the HIP side starts from bit-identical weights.
"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import transformer_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
LR = 1e-3
STEPS = 10
DECODE_SCALE = 20.0


def make(name, V, E, H, L, W, T, B, seed, full_grads):
    cfg = O.Config(V, E, W, L, H)
    params = O.init_params(V, E, W, L, seed=seed)
    rng = np.random.default_rng(seed + 100)
    for k in params:
        if k.endswith(("gamma", "beta", "bias")):
            params[k] = params[k] + 0.05 * rng.standard_normal(params[k].shape)
        params[k] = params[k].astype(np.float32).astype(np.float64)
    out = {"cfg": np.array([V, E, H, L, W, T, B], dtype=np.int64), "lr": np.float64(LR)}
    for k, v in params.items():
        out["param:" + k] = v.astype(np.float32)
    xs, ys = [], []
    for _ in range(STEPS):
        x, y = O.synthetic_batch(rng, V, B, T)
        xs.append(x); ys.append(y)
    out["x"], out["y"] = np.stack(xs), np.stack(ys)
    orc = O.OracleTransformer(cfg, params)
    loss, acc, G, logits = orc.loss_and_grads(xs[0], ys[0], training=False)
    out["logits0"] = logits.astype(np.float32)
    for k, g in G.items():
        if full_grads:
            out["grad:" + k] = g.astype(np.float32)
        out["gradnorm:" + k] = np.float64(np.sqrt((g * g).sum()))
    losses, accs = [], []
    for s in range(STEPS):
        l, a = orc.train_step(xs[s], ys[s], LR, training=False)
        losses.append(l); accs.append(a)
        if s == 2:
            for k in orc.p:
                if full_grads:
                    out["param3:" + k] = orc.p[k].astype(np.float32)
                out["param3norm:" + k] = np.float64(np.sqrt((orc.p[k] ** 2).sum()))
    out["losses"], out["accs"] = np.array(losses), np.array(accs)
    # greedy decode: sigma=0.02 random weights give a constant argmax, so the decode fixtures use the
    # SAME weights with every truncated-normal tensor multiplied by DECODE_SCALE in float32 (the test
    # rebuilds them the same way); prompt = first 10 ids of batch 0.
    kinds = {n: k for n, _, k in O.param_specs(V, E, W, L)}
    dparams = {k: ((v.astype(np.float32) * np.float32(DECODE_SCALE)) if kinds[k] == "normal"
                   else v.astype(np.float32)).astype(np.float64) for k, v in params.items()}
    out["decode_scale"] = np.float32(DECODE_SCALE)
    orc0 = O.OracleTransformer(cfg, dparams)
    prompt = xs[0][0, :10]
    n = min(32, W - 10)
    out["prompt"] = prompt.astype(np.int32)
    out["greedy_literal"] = np.array(orc0.generate_literal(prompt, n), dtype=np.int32)
    out["greedy_kv"] = np.array(orc0.generate_kv(prompt, n), dtype=np.int32)
    # top-2 margin of every kv-mode greedy decision (tells a real mismatch from a near-tie)
    ids = list(prompt)
    margins = []
    for t in out["greedy_kv"]:
        lg, _, _ = orc0.forward(np.array([ids]))
        z = np.sort(lg[0, -1])
        margins.append(z[-1] - z[-2])
        ids.append(int(t))
    out["greedy_kv_margin"] = np.array(margins)
    print(name, "min greedy margin", min(margins), "distinct", len(set(out["greedy_kv"].tolist())))
    np.savez_compressed(os.path.join(HERE, "transformer_%s.npz" % name), **out)
    print(name, "losses", np.round(losses, 5), "greedy_kv", out["greedy_kv"][:8])


if __name__ == "__main__":
    make("gA", 390, 64, 4, 2, 48, 33, 2, seed=11, full_grads=True)     # D=16, odd T
    make("gB", 390, 128, 2, 2, 80, 64, 2, seed=12, full_grads=False)   # D=64
    make("gC", 390, 64, 2, 3, 64, 40, 3, seed=13, full_grads=False)    # D=32, L=3, B=3
