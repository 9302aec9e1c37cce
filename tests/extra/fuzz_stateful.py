#!/usr/bin/env python3
"""One-off STATEFUL fuzzing against the float64 oracle (GPU box only):  python tests/extra/fuzz_stateful.py [models] [ops] [seed]
One fp32 model per round (random configuration incl. odd head sizes, dropout on), then a random sequence of operations whose
results are each compared with an oracle that is re-synchronised to the model's full state (weights, both Adam moments,
iteration count) before every operation -- so every comparison is a one-operation comparison, and anything stale inside the
product (decode state and its transposed weights, presents of an older pass, workspaces sized by an earlier shape, the pipelined
loop's staging slots, dropout streams keyed on the iteration) shows up:
  train_step | three pipelined train_step_async | loss_and_grads | evaluate | forward | forward with past (chain) |
  training=True forward | greedy generate (kv / literal) | state_dict -> perturb -> load_state_dict | set one weight.
FUZZ_DP=1: every model joins a 1-rank RCCL communicator first (the data-parallel code path of the train step)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import transformer_oracle as O
from composer_amd import _lib
from composer_amd.transformer import Transformer


def sync_oracle(orc, m):
    for n in m.parameter_names:
        orc.p[n] = m.get_parameter(n).astype(np.float64)
        orc.m[n] = m.get_parameter(n, _lib.KIND_ADAM_M).astype(np.float64)
        orc.v[n] = m.get_parameter(n, _lib.KIND_ADAM_V).astype(np.float64)
    orc.iterations = m.iterations


def close(a, b, tol, what):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    e = np.abs(a - b).max() / max(1e-30, max(1.0, np.abs(b).max()))
    assert e <= tol, (what, e)


def run_model(rng, nops, log):
    while True:
        Dl = int(rng.choice([4, 8, 12, 16, 24, 32, 40, 64]))
        H = int(rng.choice([1, 2, 3, 4]))
        E = Dl * H
        if E % 8 == 0 and E <= 192:
            break
    V = int(rng.choice([7, 390, 700])); L = int(rng.integers(1, 3)); W = int(rng.integers(6, 48)); maxB = int(rng.integers(1, 5))
    p = float(rng.choice([0.0, 0.1]))
    cfg = dict(V=V, E=E, H=H, L=L, W=W, maxB=maxB, p=p)
    log.append(("model", cfg))
    params = {k: v.astype(np.float32) for k, v in O.init_params(V, E, W, L, seed=int(rng.integers(0, 1000)), stddev=0.05).items()}
    ocfg = O.Config(V, E, W, L, H, attention_dropout_rate=p, residual_dropout_rate=p)
    bf = os.environ.get("FUZZ_DTYPE", "fp32") == "bf16"      # bf16: against the oracle that rounds where the bf16 kernels round
    orc = O.OracleTransformer(ocfg, params, seed=11, emulate_bf16=bf)
    m = Transformer(V, E, W, L, H, attention_dropout_rate=p, residual_dropout_rate=p, dtype="bf16" if bf else "fp32", seed=11, max_batch=maxB, max_seq=W)
    k = 700.0 if bf else 1.0                                 # tolerance scale
    m.set_weights(params)
    if os.environ.get("FUZZ_DP"):           # the product's data-parallel path with a 1-rank communicator (buckets, side stream, RCCL, 1/N)
        m.init_data_parallel(0, 1, Transformer.new_unique_id())
    lr = 1e-3
    try:
        for _ in range(nops):
            op = str(rng.choice(["train", "train", "async", "grads", "eval", "fwd", "fwd_kw", "past", "fwd_train", "gen_kv", "gen_lit", "reload", "setw", "bad"]))
            B = int(rng.integers(1, maxB + 1)); T = int(rng.integers(1, W + 1))
            x, y = O.synthetic_batch(rng, V, B, T)
            log.append((op, B, T))
            sync_oracle(orc, m)
            if op == "train":
                lo, ao = orc.train_step(x, y, lr, training=p > 0)
                lm, am = m.train_step(x, y, lr)
                close(lm, lo, 3e-5 * k, "train loss"); assert bf or abs(am - ao) < 1e-6
                # (bf16: Adam's first steps move every element by ~lr * sign(g), so an element whose gradient is within the bf16 error
                # of zero lands 2 * lr away -- the parameters are compared in fp32 only)
                for n in ([] if bf else m.parameter_names):
                    close(m.get_parameter(n), orc.p[n], 3e-5, "param " + n)
                assert m.iterations == orc.iterations
            elif op == "async":
                batches = [O.synthetic_batch(rng, V, B, T) for _ in range(3)]
                tickets = [m.train_step_async(bx, by, lr) for bx, by in batches]
                want = [orc.train_step(bx, by, lr, training=p > 0) for bx, by in batches]
                for tk, (lo, ao) in zip(tickets, want):
                    lm, am = m.step_metrics(tk)
                    close(lm, lo, 1e-4 * k / 3, "async loss")
                for n in ([] if bf else m.parameter_names):
                    close(m.get_parameter(n), orc.p[n], 1e-4, "async param " + n)
            elif op == "grads":
                lo, ao, G, _ = orc.loss_and_grads(x, y, training=p > 0, step=orc.iterations)
                lm, am = m.loss_and_grads(x, y)
                close(lm, lo, 3e-5 * k, "grads loss")
                worst = max(np.abs(m.get_parameter(n, _lib.KIND_GRAD) - G[n]).max() / (np.abs(G[n]).max() + 1e-5) for n in m.parameter_names)
                assert worst < (5e-2 if bf else 1e-3), ("grad", worst)
            elif op == "eval":
                lm, am = m.evaluate([(x, y)])
                lo, ao = orc.loss_acc(orc.forward(x)[0], y)
                close(lm, lo, 3e-5 * k, "eval loss"); assert bf or abs(am - ao) < 1e-6
            elif op == "fwd":
                lg, pres = m(x)
                want, opast, _ = orc.forward(x)
                close(lg, want, 2e-4 * k / 4, "logits")
                close(np.array(pres[L - 1]), opast[L - 1], 3e-5 * k, "presents")
            elif op == "fwd_kw":
                # the rest of Transformer.call's signature: position / token-type ids, attention mask (key 0 always kept: a fully
                # masked row is held to a looser bound in tests/test_gpu_model.py), with and without the train-step dropout
                kw = {}
                if rng.random() < 0.6:
                    kw["position_ids"] = rng.integers(0, W, size=(B if rng.random() < 0.5 else 1, T)).astype(np.int32)
                if rng.random() < 0.5:
                    kw["token_type_ids"] = rng.integers(0, min(V, 2), size=(B, T)).astype(np.int32)
                if rng.random() < 0.6:
                    am = (rng.random((B, T)) > 0.3).astype(np.int32); am[:, 0] = 1
                    kw["attention_mask"] = am
                tr = bool(rng.random() < 0.4)
                lg, pres = m(x, training=tr, **kw)
                want, opast, _ = orc.forward(x, training=tr and p > 0, step=orc.iterations, **kw)
                close(lg, want, 2e-4 * k / 4, "logits with " + ",".join(sorted(kw)))
                close(np.array(pres[L - 1]), opast[L - 1], 3e-5 * k, "presents")
            elif op == "past":
                if T == W:
                    continue
                lg, pres = m(x)
                _, opast, _ = orc.forward(x)
                cur = x
                for _k in range(min(3, W - T)):
                    nxt = rng.integers(0, V, size=(B, 1)).astype(np.int32)
                    cur = np.concatenate([cur, nxt], axis=1)
                    lg, pres = m(cur, past=pres)
                    want, opast, _ = orc.forward(nxt, past=opast)
                    close(lg, want, 2e-4 * k / 4, "past logits")
            elif op == "fwd_train":
                lg, _ = m(x, training=True)
                want, _, _ = orc.forward(x, training=p > 0, step=orc.iterations)
                close(lg, want, 2e-4 * k / 4, "training logits")
            elif op in ("gen_kv", "gen_lit"):
                Pn = int(rng.integers(1, W + 1)); prompt = rng.integers(0, V, size=Pn).astype(np.int32)
                n = int(rng.integers(1, 6))
                if op == "gen_kv":
                    n = min(n, W - Pn + 1)
                    if n <= 0:
                        continue
                    got = m.generate(prompt, n, temperature=0.0, mode="kv").tolist(); want = list(orc.generate_kv(prompt, n))
                else:
                    got = m.generate(prompt, n, temperature=0.0, mode="literal").tolist(); want = list(orc.generate_literal(prompt, n))
                z = orc.forward(prompt[None])[0][0, -1]
                top = np.sort(z)[-2:]
                if V > 1 and top[1] - top[0] > (1e-3 if not bf else 0.2):      # decode runs on the fp32 master weights in either mode
                    assert got[0] == want[0], (op, got, want)
            elif op == "reload":
                sd = m.state_dict()
                m.train_step(x, y, lr)                       # move away
                m.load_state_dict(sd)
                sync_oracle(orc, m)
                for n in m.parameter_names:
                    assert np.array_equal(m.get_parameter(n), sd["model/" + n]), n
                assert m.iterations == int(sd["optimizer/iter"])
            elif op == "bad":
                # a refused call must leave the model exactly as it was (the operations after it are checked as usual)
                kind = int(rng.integers(0, 6))
                before = m.iterations
                try:
                    if kind == 0:
                        bx = x.copy(); bx[0, 0] = V; m.train_step(bx, y, lr)
                    elif kind == 1:
                        by = y.copy(); by[-1, -1] = -1; m.train_step_async(x, by, lr)
                    elif kind == 2:
                        m(np.zeros((1, W + 1), np.int32))
                    elif kind == 3:
                        m.generate(np.zeros(W, np.int32), 2, temperature=0.0, mode="kv")
                    elif kind == 4:
                        m(x, past=[np.zeros((2, B, H, 3, E // H + 1), np.float32)] * L)
                    else:
                        m.evaluate([(np.full((1, 2), V + 5, np.int32), np.zeros((1, 2), np.int32))])
                    raise AssertionError("bad call %d was accepted" % kind)
                except (ValueError, IndexError, RuntimeError) as e:
                    if isinstance(e, AssertionError):
                        raise
                assert m.iterations == before
            elif op == "setw":
                n = str(rng.choice(m.parameter_names))
                w = m.get_parameter(n)
                w2 = (w + 0.01 * rng.standard_normal(w.shape)).astype(np.float32)
                m.set_parameter(n, w2)
                assert np.array_equal(m.get_parameter(n), w2)
    finally:
        m.close()


def main():
    models = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    nops = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    _lib.require_gpu()
    bad = 0
    done = {}
    for i in range(models):
        log = []
        try:
            run_model(rng, nops, log)
        except Exception as e:
            bad += 1
            print("FAIL model %d: %s %s" % (i, type(e).__name__, str(e)[:300]))
            print("   trail:", log[:1], "...", log[-6:], flush=True)
        for e in log[1:]:
            done[e[0]] = done.get(e[0], 0) + 1
    print("operations run:", dict(sorted(done.items())))
    print("%d models x %d operations, %d failures" % (models, nops, bad))
    sys.exit(min(bad, 100))


if __name__ == "__main__":
    main()
