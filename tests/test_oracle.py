"""CPU tests that pin the numpy oracle: (1) against an INDEPENDENT torch-autograd restatement of
transformer.py:696-833 (float64, must agree to ~1e-10), (2) against the committed golden vectors."""
import math
import os
import numpy as np
import pytest
import torch

from oracle import transformer_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


from oracle.torch_restatement import forward as torch_forward, TorchTrainer   # independent restatement (no shared code)


@pytest.mark.parametrize("E,H,L,T,use_ln", [(32, 4, 2, 16, True), (64, 4, 2, 33, True), (32, 2, 1, 8, False)])
def test_oracle_matches_torch_autograd(E, H, L, T, use_ln):
    V, W, B = 390, 40, 2
    cfg = O.Config(V, E, W, L, H, use_layer_normalization=use_ln)
    params = O.init_params(V, E, W, L, seed=3)
    rng = np.random.default_rng(5)
    # non-trivial gamma/beta/bias so their gradients are exercised
    for k in params:
        if k.endswith(("gamma", "beta", "bias")):
            params[k] = params[k] + 0.1 * rng.standard_normal(params[k].shape)
    x, y = O.synthetic_batch(rng, V, B, T)
    orc = O.OracleTransformer(cfg, params)
    loss, acc, G, logits = orc.loss_and_grads(x, y, training=False)

    P = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in params.items()}
    tl = torch_forward(P, torch.tensor(x, dtype=torch.long), cfg)
    tloss = torch.nn.functional.cross_entropy(tl.reshape(-1, V), torch.tensor(y, dtype=torch.long).reshape(-1))
    tloss.backward()
    assert np.allclose(logits, tl.detach().numpy(), atol=1e-11, rtol=0)
    assert abs(loss - tloss.item()) < 1e-12
    for k in params:
        if not use_ln and ("ln_1" in k or "ln_2" in k):
            continue
        g = P[k].grad.numpy()
        assert np.allclose(G[k], g, atol=1e-12, rtol=1e-9), k


def test_torch_trainer_follows_the_oracle_train_steps():
    """The torch restatement's full train step (autograd + Keras Adam), which bench.py times as the CPU baseline, tracks
    the numpy oracle's for 3 steps in float64."""
    V, E, H, L, W, T, B = 390, 32, 4, 2, 24, 20, 2
    cfg = O.Config(V, E, W, L, H)
    params = O.init_params(V, E, W, L, seed=6)
    rng = np.random.default_rng(7)
    orc = O.OracleTransformer(cfg, params)
    tt = TorchTrainer(cfg, params, dtype=torch.float64)
    for s in range(3):
        x, y = O.synthetic_batch(rng, V, B, T)
        lo, ao = orc.train_step(x, y, 1e-3, training=False)
        lt, at = tt.train_step(x, y, 1e-3)
        assert abs(lo - lt) < 1e-10 and abs(ao - at) < 1e-12
    for k in params:
        assert np.allclose(orc.p[k], tt.P[k].detach().numpy(), atol=1e-12), k


def test_adam_is_keras_formulation():
    """theta -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+1e-7): differs from torch.optim.Adam's eps placement."""
    cfg = O.Config(390, 32, 16, 1, 4)
    params = O.init_params(390, 32, 16, 1, seed=0)
    orc = O.OracleTransformer(cfg, params)
    G = {k: np.full_like(v, 0.5) for k, v in params.items()}
    p0 = {k: v.copy() for k, v in orc.p.items()}
    orc.adam_step(G, 1e-3)
    m, v = 0.05, 0.00025
    alpha = 1e-3 * math.sqrt(1 - 0.999) / (1 - 0.9)
    exp = alpha * m / (math.sqrt(v) + 1e-7)
    for k in params:
        assert np.allclose(p0[k] - orc.p[k], exp, rtol=1e-12)
    assert orc.iterations == 1


def test_kv_cache_equals_full_forward():
    """model(x, past=presents) (transformer.py:735-765) must equal the last row of the full forward."""
    cfg = O.Config(390, 32, 24, 2, 4)
    orc = O.OracleTransformer(cfg, O.init_params(390, 32, 24, 2, seed=1))
    rng = np.random.default_rng(0)
    ids = rng.integers(0, 390, size=(1, 12))
    full, _, _ = orc.forward(ids)
    lg, past, _ = orc.forward(ids[:, :7])
    for t in range(7, 12):
        lg, past, _ = orc.forward(ids[:, t:t + 1], past=past)
        assert np.allclose(lg[0, -1], full[0, t], atol=1e-12)


def test_masked_scores_are_exactly_zero_probability():
    """-1e4 masking (transformer.py:354) underflows to exactly 0 => skipping masked tiles is exact."""
    cfg = O.Config(390, 32, 16, 1, 4)
    orc = O.OracleTransformer(cfg, O.init_params(390, 32, 16, 1, seed=2))
    x = np.random.default_rng(1).integers(0, 390, size=(1, 16))
    _, _, cache = orc.forward(x, keep_cache=True)
    pr = cache["layers"][0]["pr"]
    assert (np.triu(pr[0, 0], 1) == 0).all()


def test_dropout_keep_rate_and_determinism():
    # flat sites: [tokens, features] mask -- right rate, deterministic, stream-dependent, no row/column structure
    k = O.dropout_keep_rows(123, O.dropout_stream(1, 2, 3), 512, 512, 0.1)
    assert abs(k.mean() - 0.9) < 0.005
    assert (k == O.dropout_keep_rows(123, O.dropout_stream(1, 2, 3), 512, 512, 0.1)).all()
    assert (k != O.dropout_keep_rows(123, O.dropout_stream(1, 2, 2), 512, 512, 0.1)).mean() > 0.1
    assert np.abs(k.mean(axis=0) - 0.9).max() < 0.06 and np.abs(k.mean(axis=1) - 0.9).max() < 0.06
    assert abs(np.corrcoef(k[:, :-1].ravel(), k[:, 1:].ravel())[0, 1]) < 0.01
    assert abs(np.corrcoef(k[:-1, :].ravel(), k[1:, :].ravel())[0, 1]) < 0.01
    for p in (0.25, 0.5):
        assert abs(O.dropout_keep_rows(9, 77, 256, 768, p).mean() - (1 - p)) < 0.01
    # attention mask: right rate, no row/column structure
    a = O.dropout_keep_attn(7, O.dropout_stream(0, 1, 1), 8, 256, 0.1)
    assert abs(a.mean() - 0.9) < 0.005
    assert np.abs(a.mean(axis=(0, 1)) - 0.9).max() < 0.05 and np.abs(a.mean(axis=(0, 2)) - 0.9).max() < 0.05
    assert abs(np.corrcoef(a[:, :, :-1].ravel(), a[:, :, 1:].ravel())[0, 1]) < 0.01
    assert abs(np.corrcoef(a[:, :-1, :].ravel(), a[:, 1:, :].ravel())[0, 1]) < 0.01


def test_param_count_matches_survey():
    """C2 = 19 639 296, C4 = 86 928 384, C1(W=1024) = 6 680 576."""
    n = lambda V, E, W, L: sum(int(np.prod(s)) for _, s, _ in O.param_specs(V, E, W, L))
    assert n(390, 512, 1024, 6) == 19639296
    assert n(390, 768, 2048, 12) == 86928384
    assert n(390, 256, 1024, 8) == 6680576


def load_golden(name):
    g = np.load(os.path.join(HERE, "golden", "transformer_%s.npz" % name))
    V, E, H, L, W, T, B = [int(v) for v in g["cfg"]]
    params = {k[6:]: g[k].astype(np.float64) for k in g.files if k.startswith("param:")}
    return g, O.Config(V, E, W, L, H), params


def decode_params(g, cfg, params):
    kinds = {n: k for n, _, k in O.param_specs(cfg.V, cfg.E, cfg.W, cfg.L)}
    return {k: ((v.astype(np.float32) * g["decode_scale"]) if kinds[k] == "normal"
                else v.astype(np.float32)).astype(np.float64) for k, v in params.items()}


@pytest.mark.parametrize("name", ["gA", "gB", "gC"])
def test_oracle_reproduces_golden(name):
    g, cfg, params = load_golden(name)
    orc = O.OracleTransformer(cfg, params)
    loss, acc, G, logits = orc.loss_and_grads(g["x"][0], g["y"][0], training=False)
    assert np.allclose(logits, g["logits0"], atol=1e-6)
    for k in params:
        assert abs(np.sqrt((G[k] ** 2).sum()) - g["gradnorm:" + k]) < 1e-12
    for s in range(len(g["losses"])):
        l, a = orc.train_step(g["x"][s], g["y"][s], float(g["lr"]), training=False)
        assert abs(l - g["losses"][s]) < 1e-12 and abs(a - g["accs"][s]) < 1e-12
    orc0 = O.OracleTransformer(cfg, decode_params(g, cfg, params))
    n = len(g["greedy_kv"])
    assert orc0.generate_kv(g["prompt"], n) == g["greedy_kv"].tolist()
    assert orc0.generate_literal(g["prompt"], n) == g["greedy_literal"].tolist()


def test_bf16_emulation_rounds_to_nearest_even_and_keeps_the_formulas():
    """round_bf16 against known values; the emulate_bf16 code path with rounding switched off reproduces the plain oracle's
    loss and gradients (so the emulation differs from the reference's arithmetic by the roundings only)."""
    r = O.round_bf16(np.array([1.0, 1.00390625, 1.01171875, -3.140625, 1e-30, 65504.0]))
    assert r[0] == 1.0 and r[1] == 1.0 and r[2] == 1.015625 and r[3] == -3.140625 and r[5] == 65536.0
    assert O.round_bf16(np.float32(1.005859375)) == np.float32(1.0078125)
    V, E, H, L, W, T, B = 390, 32, 4, 2, 24, 20, 2
    params = O.init_params(V, E, W, L, seed=4, stddev=0.2)
    x, y = O.synthetic_batch(np.random.default_rng(1), V, B, T)
    cfg = O.Config(V, E, W, L, H, attention_dropout_rate=0.1, residual_dropout_rate=0.1)
    a = O.OracleTransformer(cfg, params, seed=3)
    b = O.OracleTransformer(cfg, params, seed=3, emulate_bf16=True)
    b.R = lambda t: t
    la, _, Ga, _ = a.loss_and_grads(x, y, training=True, step=2)
    lb, _, Gb, _ = b.loss_and_grads(x, y, training=True, step=2)
    assert abs(la - lb) < 1e-12
    for k in Ga:
        assert np.abs(Ga[k] - Gb[k]).max() <= 1e-12 * (np.abs(Ga[k]).max() + 1e-30), k
    c = O.OracleTransformer(cfg, params, seed=3, emulate_bf16=True)
    lc, _, Gc, _ = c.loss_and_grads(x, y, training=True, step=2)
    assert 1e-6 < abs(lc - la) < 5e-2 * la


# ---------------------------------------------------------------------------------------------- oracle/run_tf_reference.py
def test_run_tf_reference_names_every_parameter_of_the_goldens():
    """The script that would pin the oracle to TensorFlow assigns the goldens' weights through the reference's object graph: its
    name -> attribute-path table must cover exactly the goldens' parameter names (no TensorFlow needed for this part)."""
    from oracle import run_tf_reference as R
    for name in ("gA", "gB", "gC"):
        g, cfg, params = load_golden(name)
        assert set(R.reference_variable_paths(cfg.L)) == set(params), name
        assert set(R.golden_params(g)) == set(params)


@pytest.mark.skipif(not os.path.isdir("/root/reference/composer"), reason="the reference tree exists in the build container only")
def test_run_tf_reference_attribute_paths_exist_in_the_reference_source():
    """Every attribute the script dereferences on the reference model is assigned in the reference's source: `self.<attr> =` in the
    class it is looked up on (transformer.py:117,189-190,257-270,482-495,551-571,673-694) -- read with ast, nothing is imported."""
    import ast
    from oracle import run_tf_reference as R
    tree = ast.parse(open("/root/reference/composer/models/transformer.py").read())
    assigned = {}
    for cls in [n for n in ast.walk(tree) if isinstance(n, ast.ClassDef)]:
        attrs = set()
        for n in ast.walk(cls):
            if isinstance(n, ast.Attribute) and isinstance(n.ctx, ast.Store) and isinstance(n.value, ast.Name) and n.value.id == "self":
                attrs.add(n.attr)
        assigned[cls.name] = attrs
    owner = {"wte": "Transformer", "wpe": "Transformer", "ln_f": "Transformer", "decoder_blocks": "Transformer",
             "ln_1": "DecoderBlock", "ln_2": "DecoderBlock", "attn": "DecoderBlock", "mlp": "DecoderBlock",
             "c_attn": "Attention", "c_fc": "MultilayerPerceptron"}
    for path in R.reference_variable_paths(2).values():
        for i, step in enumerate(path[:-1]):
            if isinstance(step, int):
                continue
            if step == "c_proj":
                cls = "Attention" if "attn" in path else "MultilayerPerceptron"
            else:
                cls = owner[step]
            assert step in assigned[cls], (step, cls)
        leaf, parent = path[-1], path[-2]
        if parent in ("c_attn", "c_proj", "c_fc"):
            assert leaf in assigned["Conv1D"], leaf                      # weight, bias (Conv1D.build)
        elif parent == "wte":
            assert leaf in assigned["SharedTokenEmbedding"], leaf        # weight
        else:
            assert leaf in ("gamma", "beta", "embeddings")               # Keras LayerNormalization / Embedding variables


def test_run_tf_reference_reports_unpinned_without_tensorflow_or_pins_with_it(tmp_path):
    """Exit status 3 = TensorFlow not importable (this container, the GPU box): parity stays "unpinned".  With TensorFlow AND the
    reference tree present the same call runs every comparison and must return 0."""
    from oracle import run_tf_reference as R
    try:
        import tensorflow  # noqa: F401
        have_tf = True
    except Exception:
        have_tf = False
    rc = R.main(["--out", str(tmp_path / "tf"), "--golden", "gA"])
    if have_tf and os.path.isdir("/root/reference/composer"):
        assert rc == 0
    else:
        assert rc == 3
