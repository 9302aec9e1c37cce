#!/usr/bin/env python3
"""One-off fuzzing of the model configuration domain against the float64 oracle (test infrastructure, GPU box only):
    python tests/extra/fuzz_shapes.py [N] [seed]
Every case draws vocabulary, width, heads (any head size up to 128), blocks, window, T, B, the constructor switches, dropout and
dtype, then compares one training-mode loss + all gradients, the inference logits, presents and a short greedy decode (fp32) with
the oracle.  Prints the failing configurations; exit code = number of failures.  The committed tests hold seeded samples of
the same generator's ranges (tests/test_gpu_round3.py)."""
import os, sys, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import transformer_oracle as O
from composer_amd import _lib
from composer_amd.transformer import Transformer


def draw(rng):
    while True:
        Dl = int(rng.integers(1, 33)) * 4 if rng.random() < 0.7 else int(rng.choice([16, 32, 64, 128]))
        H = int(rng.choice([1, 2, 3, 4, 5, 6, 8]))
        E = Dl * H
        wide = os.environ.get("FUZZ_WIDE")               # FUZZ_WIDE=1: widths up to 1024 and windows up to 400
        if E % 8 or E > (1024 if wide else 512) or Dl > 128:
            continue
        L = int(rng.integers(1, 4))
        W = int(rng.integers(1, 400 if wide and rng.random() < 0.3 else 90))
        T = int(rng.integers(1, W + 1))
        B = int(rng.integers(1, 7))
        V = int(rng.choice([2, 3, 17, 390, 513, 1000, 2500]))
        return dict(V=V, E=E, H=H, L=L, W=W, T=T, B=B, scale=bool(rng.integers(0, 2)), use_ln=bool(rng.integers(0, 4) > 0),
                    eps=float(rng.choice([1e-5, 1e-3])), p=float(rng.choice([0.0, 0.1, 0.3])), dtype=str(rng.choice(["fp32", "bf16"])),
                    seed=int(rng.integers(0, 1 << 20)))


def check(c):
    V, E, H, L, W, T, B = c["V"], c["E"], c["H"], c["L"], c["W"], c["T"], c["B"]
    rng = np.random.default_rng(c["seed"])
    std = float(os.environ.get("FUZZ_STD", "0.1"))          # 0.1 with no LayerNorm / no scaling at E >= 256 is numerically explosive
    params = {k: v.astype(np.float32) for k, v in O.init_params(V, E, W, L, seed=c["seed"] % 1000, stddev=std).items()}
    for k in params:
        if k.endswith(("gamma", "beta", "bias")):
            params[k] = (params[k] + 0.05 * rng.standard_normal(params[k].shape)).astype(np.float32)
    x, y = O.synthetic_batch(rng, V, B, T)
    ocfg = O.Config(V, E, W, L, H, layer_normalization_epsilon=c["eps"], scale=c["scale"], use_layer_normalization=c["use_ln"],
                    attention_dropout_rate=c["p"], residual_dropout_rate=c["p"])
    bf = c["dtype"] == "bf16"
    orc = O.OracleTransformer(ocfg, params, seed=7, emulate_bf16=bf)
    m = Transformer(V, E, W, L, H, attention_dropout_rate=c["p"], residual_dropout_rate=c["p"], layer_normalization_epsilon=c["eps"],
                    scale=c["scale"], use_layer_normalization=c["use_ln"], dtype=c["dtype"], seed=7, max_batch=B, max_seq=W)
    try:
        m.set_weights(params)
        loss, acc, G, _ = orc.loss_and_grads(x, y, training=c["p"] > 0, step=0)
        l2, a2 = m.loss_and_grads(x, y)
        assert abs(l2 - loss) <= (3e-5 if not bf else 2e-2) * abs(loss), ("loss", l2, loss)
        worst = 0.0
        # bf16: the floor also scales with the model's largest gradient (1e-3 of it) -- with a two-word vocabulary ln_f/beta cancels to
        # ~2e-5 and carries the bf16 rounding residue of its summands (seed 79 of round 5: 1.3e-6 absolute = 6 % of that tensor's
        # maximum, bit-identical on the round-4 library)
        gfloor = 1e-5 + (1e-3 * max(np.abs(G[n]).max() for n in m.parameter_names) if bf else 0.0)
        for n in m.parameter_names:
            if not c["use_ln"] and ("ln_1" in n or "ln_2" in n):
                continue
            gr = m.get_parameter(n, _lib.KIND_GRAD).astype(np.float64)
            # relative to the tensor's largest element, with a floor: a gradient that cancels to ~1e-7 (ln_f/beta with a two-word
            # vocabulary) carries fp32 rounding residue of the same size
            worst = max(worst, np.abs(gr - G[n]).max() / (np.abs(G[n]).max() + gfloor))
        assert worst <= (1e-3 if not bf else 5e-2), ("grad", worst)
        lg, pres = m(x)
        want, opast, _ = orc.forward(x)
        assert np.abs(lg - want).max() <= (2e-4 if not bf else 4e-2) * max(1.0, np.abs(want).max()), "logits"
        if not bf:
            for i in range(L):
                assert np.abs(np.array(pres[i]) - opast[i]).max() <= 3e-5 * max(1.0, np.abs(opast[i]).max()), "presents"
            n = min(4, W - T + 1)
            if n > 0:
                prompt = x[0, :T]
                z = orc.forward(prompt[None])[0][0, -1]
                top = np.sort(z)[-2:]
                if V > 1 and top[1] - top[0] > 1e-3:          # first step is not a near-tie
                    got = m.generate(prompt, n, temperature=0.0, mode="kv").tolist()
                    wantids = list(orc.generate_kv(prompt, n))
                    assert got[0] == wantids[0], ("decode", got, wantids)
    finally:
        m.close()


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    _lib.require_gpu()
    bad = 0
    for i in range(N):
        c = draw(rng)
        try:
            check(c)
        except Exception as e:
            bad += 1
            print("FAIL", c, "->", type(e).__name__, str(e)[:300], flush=True)
    print("%d cases, %d failures" % (N, bad))
    sys.exit(min(bad, 100))


if __name__ == "__main__":
    main()
