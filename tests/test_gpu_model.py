"""-m gpu: end-to-end parity of the HIP train/eval/decode path (through the host-side Transformer class and the
C ABI) against the committed golden vectors of the float64 oracle (tests/golden/*.npz) and against the oracle
itself on seeded inputs (dropout on, using the shared counter-hash masks).

Tolerances: fp32 mode -- logits max-abs <= 1e-4, 10-step loss relative <= 1e-4, greedy ids
identical; bf16 mode -- 10-step loss relative <= 2e-2.
"""
import os
import numpy as np
import pytest

from oracle import transformer_oracle as O

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def load_golden(name):
    g = np.load(os.path.join(HERE, "golden", "transformer_%s.npz" % name))
    V, E, H, L, W, T, B = [int(v) for v in g["cfg"]]
    params = {k[6:]: g[k] for k in g.files if k.startswith("param:")}
    return g, (V, E, H, L, W, T, B), params


def make_model(cfg, params, dtype, p_attn=0.0, p_resid=0.0, seed=0, use_ln=True, max_batch=None):
    from composer_amd.transformer import Transformer
    V, E, H, L, W, T, B = cfg
    m = Transformer(V, E, W, L, H, attention_dropout_rate=p_attn, residual_dropout_rate=p_resid,
                    use_layer_normalization=use_ln, dtype=dtype, seed=seed, max_batch=max_batch or B, max_seq=W)
    m.set_weights(params)
    return m


@pytest.mark.parametrize("name", ["gA", "gB", "gC"])
def test_fp32_forward_and_gradients_match_golden(name):
    g, cfg, params = load_golden(name)
    m = make_model(cfg, params, "fp32")
    logits, _ = m(g["x"][0])
    assert np.abs(logits - g["logits0"]).max() <= 1e-4
    loss, acc = m.loss_and_grads(g["x"][0], g["y"][0])
    assert abs(loss - g["losses"][0]) <= 1e-5 * abs(g["losses"][0])
    assert abs(acc - g["accs"][0]) < 1e-6
    from composer_amd import _lib
    for n in m.parameter_names:
        gr = m.get_parameter(n, _lib.KIND_GRAD).astype(np.float64)
        nr = float(g["gradnorm:" + n])
        assert abs(np.sqrt((gr ** 2).sum()) - nr) <= 2e-4 * nr + 1e-9, n
        if "grad:" + n in g.files:
            assert np.abs(gr - g["grad:" + n]).max() <= 2e-4 * np.abs(g["grad:" + n]).max() + 1e-9, n
    m.close()


@pytest.mark.parametrize("name", ["gA", "gB", "gC"])
def test_fp32_ten_step_loss_curve_matches_golden(name):
    g, cfg, params = load_golden(name)
    m = make_model(cfg, params, "fp32")
    for s in range(len(g["losses"])):
        loss, acc = m.train_step(g["x"][s], g["y"][s], float(g["lr"]))
        assert abs(loss - g["losses"][s]) <= 1e-4 * abs(g["losses"][s]), (s, loss, g["losses"][s])
        assert abs(acc - g["accs"][s]) < 1e-6
        if s == 2:
            for n in m.parameter_names:
                if "param3:" + n in g.files:
                    assert np.abs(m.get_parameter(n) - g["param3:" + n]).max() <= 2e-5, n
                pn = float(g["param3norm:" + n])
                assert abs(np.sqrt((m.get_parameter(n).astype(np.float64) ** 2).sum()) - pn) <= 1e-4 * pn + 1e-9, n
    assert m.iterations == len(g["losses"])
    m.close()


@pytest.mark.parametrize("name", ["gA", "gB", "gC"])
def test_bf16_ten_step_loss_curve_within_tolerance(name):
    g, cfg, params = load_golden(name)
    m = make_model(cfg, params, "bf16")
    for s in range(len(g["losses"])):
        loss, acc = m.train_step(g["x"][s], g["y"][s], float(g["lr"]))
        assert abs(loss - g["losses"][s]) <= 2e-2 * abs(g["losses"][s]), (s, loss, g["losses"][s])
    m.close()


def decode_params(g, cfg, params):
    V, E, H, L, W, T, B = cfg
    kinds = {n: k for n, _, k in O.param_specs(V, E, W, L)}
    return {k: (v.astype(np.float32) * g["decode_scale"]) if kinds[k] == "normal" else v.astype(np.float32)
            for k, v in params.items()}


@pytest.mark.parametrize("name", ["gA", "gB", "gC"])
@pytest.mark.parametrize("graph", [True, False])
def test_fp32_greedy_decode_is_bit_exact(name, graph, monkeypatch):
    g, cfg, params = load_golden(name)
    monkeypatch.setenv("COMPOSER_NO_GRAPH", "0" if graph else "1")
    m = make_model(cfg, decode_params(g, cfg, params), "fp32")
    n = len(g["greedy_kv"])
    kv = m.generate(g["prompt"], n, temperature=0.0, mode="kv")
    assert kv.tolist() == g["greedy_kv"].tolist(), (kv.tolist(), g["greedy_kv"].tolist(), g["greedy_kv_margin"].min())
    lit = m.generate(g["prompt"], n, temperature=0.0, mode="literal")
    assert lit.tolist() == g["greedy_literal"].tolist()
    # continuing a decode returns the same ids as one long call
    m.generate(g["prompt"], 1, temperature=0.0, mode="kv")
    m.close()


def test_decode_window_overflow_is_an_error():
    g, cfg, params = load_golden("gA")
    m = make_model(cfg, params, "fp32")
    W = cfg[4]
    with pytest.raises(IndexError):
        m.generate(g["prompt"], W, temperature=0.0, mode="kv")     # 10 + W - 1 > W
    out = m.generate(g["prompt"], W - 9, temperature=0.0, mode="kv")   # exactly fills the window
    assert len(out) == W - 9
    m.close()


def test_sampling_is_seeded_and_varies():
    g, cfg, params = load_golden("gA")
    m = make_model(cfg, decode_params(g, cfg, params), "fp32")
    a = m.generate(g["prompt"], 24, temperature=1.0, mode="kv", seed=1)
    b = m.generate(g["prompt"], 24, temperature=1.0, mode="kv", seed=1)
    c = m.generate(g["prompt"], 24, temperature=1.0, mode="kv", seed=2)
    assert a.tolist() == b.tolist() and a.tolist() != c.tolist()
    assert ((a >= 0) & (a < cfg[0])).all()
    m.close()


@pytest.mark.parametrize("dtype,tol", [("fp32", 2e-4), ("bf16", 3e-2)])
def test_dropout_training_step_matches_oracle(dtype, tol):
    """Dropout ON (p=0.1 as in default_config.yml): both sides draw masks from the same counter hash, so the
    training-mode loss and gradients are comparable exactly."""
    V, E, H, L, W, T, B = 390, 64, 4, 2, 48, 40, 2
    params = {k: v.astype(np.float32) for k, v in O.init_params(V, E, W, L, seed=21).items()}
    rng = np.random.default_rng(8)
    for k in params:
        if k.endswith(("gamma", "beta", "bias")):
            params[k] = (params[k] + 0.05 * rng.standard_normal(params[k].shape)).astype(np.float32)
    x, y = O.synthetic_batch(rng, V, B, T)
    ocfg = O.Config(V, E, W, L, H, attention_dropout_rate=0.1, residual_dropout_rate=0.1)
    orc = O.OracleTransformer(ocfg, params, seed=99)
    loss, acc, G, _ = orc.loss_and_grads(x, y, training=True, step=0)
    m = make_model((V, E, H, L, W, T, B), params, dtype, p_attn=0.1, p_resid=0.1, seed=99)
    l2, a2 = m.loss_and_grads(x, y)
    assert abs(l2 - loss) <= tol * abs(loss)
    from composer_amd import _lib
    worst = 0.0
    for n in m.parameter_names:
        gr = m.get_parameter(n, _lib.KIND_GRAD).astype(np.float64)
        worst = max(worst, np.abs(gr - G[n]).max() / (np.abs(G[n]).max() + 1e-12))
    assert worst <= (5e-4 if dtype == "fp32" else 8e-2), worst
    m.close()


def test_no_layernorm_variant_matches_oracle():
    V, E, H, L, W, T, B = 390, 64, 2, 2, 32, 24, 2
    params = {k: v.astype(np.float32) for k, v in O.init_params(V, E, W, L, seed=5).items()}
    rng = np.random.default_rng(2)
    x, y = O.synthetic_batch(rng, V, B, T)
    orc = O.OracleTransformer(O.Config(V, E, W, L, H, use_layer_normalization=False), params)
    loss, acc, G, _ = orc.loss_and_grads(x, y, training=False)
    m = make_model((V, E, H, L, W, T, B), params, "fp32", use_ln=False)
    l2, _ = m.loss_and_grads(x, y)
    assert abs(l2 - loss) <= 1e-5 * abs(loss)
    from composer_amd import _lib
    for n in m.parameter_names:
        if "ln_1" in n or "ln_2" in n:
            continue
        gr = m.get_parameter(n, _lib.KIND_GRAD).astype(np.float64)
        assert np.abs(gr - G[n]).max() <= 3e-4 * np.abs(G[n]).max() + 1e-9, n
    m.close()


def test_evaluate_matches_oracle_loss():
    g, cfg, params = load_golden("gC")
    m = make_model(cfg, params, "fp32")
    ds = [(g["x"][i], g["y"][i]) for i in range(3)]
    loss, acc = m.evaluate(ds)
    orc = O.OracleTransformer(O.Config(cfg[0], cfg[1], cfg[4], cfg[3], cfg[2]), {k: v.astype(np.float64) for k, v in params.items()})
    ref = [orc.loss_acc(orc.forward(x)[0], y) for x, y in ds]
    assert abs(loss - np.mean([r[0] for r in ref])) < 1e-5 and abs(acc - np.mean([r[1] for r in ref])) < 1e-6
    m.close()


def test_single_rank_communicator_does_not_change_the_step():
    """RCCL path with a 1-rank communicator.  model.hip runs the data-parallel sequence whenever a communicator exists
    (bucket event -> side-stream ncclAllReduce per decoder block, the 3-float metrics all-reduce, Adam behind the last
    bucket with the 1/N scale), so this executes the product DP code of the 8-GPU job on one GPU; results must not move."""
    g, cfg, params = load_golden("gA")
    from composer_amd.transformer import Transformer
    m = make_model(cfg, params, "fp32")
    m.init_data_parallel(0, 1, Transformer.new_unique_id())
    assert m.all_reduce_sum([1.5, -2.0, 3.25]).tolist() == [1.5, -2.0, 3.25]          # cmp_dp_allreduce_test
    for s in range(3):
        loss, acc = m.train_step(g["x"][s], g["y"][s], float(g["lr"]))
        assert abs(loss - g["losses"][s]) <= 1e-4 * abs(g["losses"][s])
        assert abs(acc - g["accs"][s]) < 1e-6                                           # metrics went through the all-reduce
    for n in m.parameter_names:
        if "param3:" + n in g.files:
            assert np.abs(m.get_parameter(n) - g["param3:" + n]).max() <= 2e-5, n
    m.close()


def test_exposed_communication_telemetry():
    """cmp_dp_stats (SURVEY 8d "exposed comm time per step"): with a 1-rank communicator the step takes the product's
    data-parallel sequence (every bucket all-reduced and Adam-updated on the communication stream); the call reports the steps,
    the bytes handed to RCCL (all gradients in fp32 + the 3-float metrics message), the message count (L+2 buckets + 1) and how
    long the compute stream waited for the communication stream at the end of each step.  A kernel that holds the communication
    stream for 20 ms before the step (cmp_dp_test_hog) must show up as exposed time; without it the wait is far smaller."""
    from composer_amd.transformer import Transformer
    from composer_amd import _lib
    g, cfg, params = load_golden("gA")
    V, E, H, L, W, T, B = cfg
    m = make_model(cfg, params, "fp32")
    assert m.dp_stats() == {"steps": 0, "exposed_ms": 0.0, "exposed_ms_total": 0.0, "bytes": 0, "buckets": 0}      # no communicator
    m.init_data_parallel(0, 1, Transformer.new_unique_id())
    for s in range(3):
        loss, _ = m.train_step(g["x"][s], g["y"][s], float(g["lr"]))
        assert abs(loss - g["losses"][s]) <= 1e-4 * abs(g["losses"][s])
    st = m.dp_stats(reset=True)
    nparam = sum(int(np.prod(m.parameter_shape(n))) for n in m.parameter_names)
    assert st["steps"] == 3 and st["buckets"] == L + 3
    assert nparam * 4 + 12 <= st["bytes"] <= (nparam + 8 * len(m.parameter_names)) * 4 + 12        # tensors are padded to 8 elements
    assert 0.0 <= st["exposed_ms"] < 5.0
    quiet = st["exposed_ms"]
    assert m.dp_stats()["steps"] == 0
    for s in range(3, 6):
        _lib.check(_lib.load().cmp_dp_test_hog(m._ctx, 8, 20000), "cmp_dp_test_hog")              # 20 ms on the communication stream
        m.train_step(g["x"][s], g["y"][s], float(g["lr"]))
    st = m.dp_stats()
    assert st["steps"] == 3 and st["exposed_ms"] > 10.0 and st["exposed_ms"] > 10 * quiet, (st, quiet)
    # the telemetry ring is reused after 32 steps without losing a step
    m.dp_stats(reset=True)
    for s in range(40):
        m.train_step(g["x"][s % 10], g["y"][s % 10], float(g["lr"]))
    assert m.dp_stats()["steps"] == 40
    m.close()


def test_two_shard_gradient_mean_equals_global_batch_on_the_hip_path():
    """Data-parallel equivalence on the product kernels: the mean of the HIP gradients of rows [0,B) and
    [B,2B) -- what the RCCL sum + the 1/N scale in the Adam kernel produce -- equals the HIP gradient of the 2B global batch."""
    from composer_amd import _lib
    V, E, H, L, W, T, B = 390, 64, 4, 2, 40, 40, 3
    params = {k: v.astype(np.float32) for k, v in O.init_params(V, E, W, L, seed=31).items()}
    x, y = O.synthetic_batch(np.random.default_rng(5), V, 2 * B, T)
    m = make_model((V, E, H, L, W, T, 2 * B), params, "fp32")
    grads = []
    for sl in (slice(0, B), slice(B, 2 * B), slice(0, 2 * B)):
        loss, _ = m.loss_and_grads(x[sl], y[sl])
        grads.append((loss, {n: m.get_parameter(n, _lib.KIND_GRAD).astype(np.float64) for n in m.parameter_names}))
    assert abs(0.5 * (grads[0][0] + grads[1][0]) - grads[2][0]) < 1e-5
    for n in m.parameter_names:
        mean = 0.5 * (grads[0][1][n] + grads[1][1][n])
        assert np.abs(mean - grads[2][1][n]).max() <= 2e-5 * np.abs(grads[2][1][n]).max() + 1e-10, n
    m.close()


def test_gemm_cu_cap_under_a_communicator_keeps_the_results():
    """cmp_dp_set_gemm_cus: with a communicator the persistent GEMM kernels launch on a subset of the CUs (leaving the rest to
    RCCL).  The tile arithmetic does not depend on the grid, so three bf16 steps at a size that takes the persistent kernels
    give the same losses with 256, 200 and 37 workgroups."""
    from composer_amd.transformer import Transformer
    V, E, H, L, W, T, B = 390, 256, 4, 2, 256, 256, 8
    params = {k: v.astype(np.float32) for k, v in O.init_params(V, E, W, L, seed=2).items()}
    x, y = O.synthetic_batch(np.random.default_rng(9), V, B, T)
    curves = []
    for cus in (None, 200, 37):
        m = make_model((V, E, H, L, W, T, B), params, "bf16")
        m.init_data_parallel(0, 1, Transformer.new_unique_id(), gemm_cus=cus)
        curves.append([m.train_step(x, y, 1e-3)[0] for _ in range(3)])
        m.close()
    assert np.allclose(curves[0], curves[1], rtol=2e-3) and np.allclose(curves[0], curves[2], rtol=2e-3), curves
    assert curves[0][-1] < curves[0][0]


def test_call_with_past_matches_oracle():
    """Transformer.call(inputs, past=presents) (transformer.py:735-765, 423-426): feeding the presents of an earlier call back
    from the host reproduces the oracle's forward(past=...) logits, step by step, and the returned presents grow by one
    position per call.  Both accepted forms of `past`: the lazy Presents object and a list of arrays."""
    g, cfg, params = load_golden("gB")
    V, E, H, L, W, T, B = cfg
    m = make_model(cfg, params, "fp32")
    orc = O.OracleTransformer(O.Config(V, E, W, L, H), {k: v.astype(np.float64) for k, v in params.items()})
    x = g["x"][0][:, :12]
    lg, pres = m(x[:, :7])
    olg, opast, _ = orc.forward(x[:, :7])
    assert np.abs(lg - olg).max() <= 1e-4
    for t in range(7, 12):
        past = pres if t % 2 else [np.array(p) for p in pres]
        lg, pres = m(x[:, :t + 1], past=past)                      # only the last token is used (:735-737)
        olg, opast, _ = orc.forward(x[:, t:t + 1], past=opast)
        assert lg.shape == (B, 1, V)
        assert np.abs(lg - olg).max() <= 1e-4, t
        assert len(pres) == L and pres[0].shape == (2, B, H, t + 1, E // H)
        assert np.abs(pres[L - 1] - opast[L - 1]).max() <= 2e-5 * max(1.0, np.abs(opast[L - 1]).max())
    # ... and equals the last row of one full forward pass
    full, _ = m(x)
    assert np.abs(full[:, -1] - lg[:, 0]).max() <= 1e-4
    with pytest.raises(IndexError):
        m(x[:, :1], past=[np.zeros((2, B, H, W, E // H), np.float32)] * L)      # position W is outside the wpe table
    m.close()


@pytest.mark.parametrize("dtype,tol", [("fp32", 2e-5), ("bf16", 3e-2)])
def test_output_hidden_states_match_the_oracle(dtype, tol):
    """Transformer(..., output_hidden_states=True) (transformer.py:610-614, 800-816, 824-825): the call returns a third element,
    the input of every decoder block followed by the ln_f output -- L + 1 tensors [B, T, E]; with `past` they cover the one new
    position; with use_cache=False the tuple is (logits, hidden states)."""
    from composer_amd.transformer import Transformer
    V, E, H, L, W, T, B = 390, 64, 4, 3, 48, 40, 2
    params = {k: v.astype(np.float32) for k, v in O.init_params(V, E, W, L, seed=5, stddev=0.1).items()}
    x, _ = O.synthetic_batch(np.random.default_rng(3), V, B, T)
    orc = O.OracleTransformer(O.Config(V, E, W, L, H), {k: v.astype(np.float64) for k, v in params.items()},
                              emulate_bf16=(dtype == "bf16"))
    m = Transformer(V, E, W, L, H, attention_dropout_rate=0.0, residual_dropout_rate=0.0, output_hidden_states=True, dtype=dtype,
                    max_batch=B, max_seq=W)
    m.set_weights(params)

    def check(hidden, cache, t):
        assert len(hidden) == L + 1
        for i, (got, want) in enumerate(zip(hidden, cache["hidden"])):
            assert got.shape == (B, t, E) and got.dtype == np.float32
            assert np.abs(got - want).max() <= tol * max(1.0, np.abs(want).max()), i

    logits, pres, hidden = m(x[:, :30])
    olg, opast, cache = orc.forward(x[:, :30])
    check(hidden, cache, 30)
    logits, pres, hidden = m(x[:, :31], past=pres)                    # one new position
    olg, opast, cache = orc.forward(x[:, 30:31], past=opast)
    check(hidden, cache, 1)
    out = m(x, use_cache=False)
    assert len(out) == 2 and len(out[1]) == L + 1 and out[1][0].shape == (B, T, E)
    m.close()


@pytest.mark.parametrize("dtype,tol", [("fp32", 2e-5), ("bf16", 3e-2)])
def test_output_attention_weights_match_the_oracle(dtype, tol):
    """Transformer(..., output_attention_weights=True) (transformer.py:360-369, 808-809, 827-831): the call's last element holds, per
    decoder block, the attention probabilities [B, H, queries, keys] after their dropout -- rows sum to 1 without dropout, are
    exactly zero above the diagonal, follow an attention_mask, and cover the one new query against all keys with `past`."""
    from composer_amd.transformer import Transformer
    V, E, H, L, W, T, B = 390, 64, 4, 2, 48, 40, 2
    params = {k: v.astype(np.float32) for k, v in O.init_params(V, E, W, L, seed=9, stddev=0.1).items()}
    rng = np.random.default_rng(6)
    x, _ = O.synthetic_batch(rng, V, B, T)
    ocfg = O.Config(V, E, W, L, H, attention_dropout_rate=0.1, residual_dropout_rate=0.1)
    orc = O.OracleTransformer(ocfg, {k: v.astype(np.float64) for k, v in params.items()}, seed=17, emulate_bf16=(dtype == "bf16"))
    m = Transformer(V, E, W, L, H, attention_dropout_rate=0.1, residual_dropout_rate=0.1, output_hidden_states=True,
                    output_attention_weights=True, dtype=dtype, seed=17, max_batch=B, max_seq=W)
    m.set_weights(params)
    mask = (rng.random((B, T)) > 0.25).astype(np.int32)
    mask[:, 0] = 1
    for kw in (dict(), dict(training=True), dict(attention_mask=mask)):
        logits, pres, hidden, att = m(x, **kw)
        _, opast, cache = orc.forward(x, step=0, **kw)
        assert len(att) == L and len(hidden) == L + 1
        for i in range(L):
            assert att[i].shape == (B, H, T, T) and att[i].dtype == np.float32
            assert np.abs(att[i] - cache["attentions"][i]).max() <= tol, (i, sorted(kw))
            assert not np.triu(att[i], 1).any()
        if not kw.get("training"):
            assert np.abs(att[0].sum(-1) - 1.0).max() <= (1e-5 if dtype == "fp32" else 2e-2)
    logits, pres, hidden, att = m(np.concatenate([x, x[:, :1]], 1), past=pres,
                                  attention_mask=np.concatenate([mask, np.ones((B, 1), np.int32)], 1))
    _, _, cache = orc.forward(x[:, :1], past=opast, attention_mask=np.concatenate([mask, np.ones((B, 1), np.int32)], 1))
    assert att[L - 1].shape == (B, H, 1, T + 1)
    assert np.abs(att[L - 1] - cache["attentions"][L - 1]).max() <= tol
    out = m(x, use_cache=False)
    assert len(out) == 3 and len(out[2]) == L
    m.close()


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-4), ("bf16", 6e-2)])
def test_position_ids_and_token_type_ids_match_the_oracle(dtype, tol):
    """Transformer.call(position_ids=, token_type_ids=) (transformer.py:770-773, 784-793): wpe rows chosen per token ([1,T]
    broadcast over the batch, or [B,T]) and a second wte row added to every token's embedding; with `past` the last
    token-type id is used (:741-742)."""
    V, E, H, L, W, T, B = 390, 64, 4, 2, 48, 24, 3
    params = {k: v.astype(np.float32) for k, v in O.init_params(V, E, W, L, seed=6, stddev=0.1).items()}
    rng = np.random.default_rng(4)
    x, _ = O.synthetic_batch(rng, V, B, T)
    orc = O.OracleTransformer(O.Config(V, E, W, L, H), {k: v.astype(np.float64) for k, v in params.items()},
                              emulate_bf16=(dtype == "bf16"))
    m = make_model((V, E, H, L, W, T, B), params, dtype)
    close = lambda a, b: np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max())
    pos_row = rng.permutation(W)[:T][None].astype(np.int32)                     # [1, T]: shared by the batch
    pos_all = rng.integers(0, W, size=(B, T)).astype(np.int32)                  # [B, T]
    typ = rng.integers(0, 2, size=(B, T)).astype(np.int32)
    base, _ = m(x)
    for kw in (dict(position_ids=pos_row), dict(position_ids=pos_all), dict(token_type_ids=typ),
               dict(position_ids=pos_all, token_type_ids=typ)):
        got, pres = m(x, **kw)
        want, opast, _ = orc.forward(x, **kw)
        assert close(got, want), sorted(kw)
        assert not close(got, base)                                            # ... and the ids do change the result
    # one more token on top of the last call's presents: its own position id, the last token-type id
    got, _ = m(np.concatenate([x, x[:, :1]], 1), past=pres, position_ids=np.full((B, 1), 5, np.int32),
               token_type_ids=np.concatenate([typ, 1 - typ[:, :1]], 1))
    want, _, _ = orc.forward(x[:, :1], past=opast, position_ids=np.full((B, 1), 5), token_type_ids=np.concatenate([typ, 1 - typ[:, :1]], 1))
    assert close(got, want)
    again, _ = m(x)
    assert np.array_equal(again, base)                                         # the ids do not stick to the model
    with pytest.raises(IndexError):
        m(x, position_ids=np.full((1, T), W, np.int32))
    with pytest.raises(Exception):
        m(x, token_type_ids=np.full((B, T), V, np.int32))
    with pytest.raises(NotImplementedError):
        m(x, input_embeddings=np.zeros((B, T, E), np.float32))
    # a fractional attention mask is a SOFT mask in the reference (float32 cast, transformer.py:774-779); the HIP path carries
    # 0 / 1 and refuses anything else instead of truncating it to "masked"
    soft = np.ones((B, T), np.float32)
    soft[0, 1] = 0.5
    with pytest.raises(ValueError):
        m(x, attention_mask=soft)
    m(x, attention_mask=np.ones((B, T), np.float32))                            # float 0 / 1 masks are fine
    m.close()


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-4), ("bf16", 6e-2)])
@pytest.mark.parametrize("T", [24, 200])
def test_attention_mask_matches_the_oracle(dtype, tol, T):
    """Transformer.call(attention_mask=) (transformer.py:774-779, 356-358): (1 - mask) * -1e4 added to the scaled, causally masked
    scores.  Right padding (every query keeps an unmasked key), scattered holes, left padding (the padding queries' allowed keys
    are ALL masked: their softmax then runs over scores near -1e4, where the reference's fp32 sum has an ulp of 1e-3 -- those
    positions are held to 5e-3 against the float64 oracle), dropout on, and one more token on top of `past`."""
    V, E, H, L, W, B = 390, 64, 4, 2, 256, 3
    params = {k: v.astype(np.float32) for k, v in O.init_params(V, E, W, L, seed=8, stddev=0.1).items()}
    rng = np.random.default_rng(5)
    x, _ = O.synthetic_batch(rng, V, B, T)
    ocfg = O.Config(V, E, W, L, H, attention_dropout_rate=0.1, residual_dropout_rate=0.1)
    orc = O.OracleTransformer(ocfg, {k: v.astype(np.float64) for k, v in params.items()}, seed=31, emulate_bf16=(dtype == "bf16"))
    m = make_model((V, E, H, L, W, T, B), params, dtype, p_attn=0.1, p_resid=0.1, seed=31)
    err = lambda a, b: np.abs(a - b).max() / max(1.0, np.abs(b).max())
    base, _ = m(x)
    right = np.ones((B, T), np.int32)
    for b in range(B):
        right[b, T - 1 - 3 * b:] = 0 if b else 1
    holes = (rng.random((B, T)) > 0.3).astype(np.int32)
    holes[:, 0] = 1                                                             # key 0 stays: no fully masked row
    for name, mask, training in (("right", right, False), ("holes", holes, False), ("holes", holes, True)):
        got, pres = m(x, attention_mask=mask, training=training)
        want, opast, _ = orc.forward(x, attention_mask=mask, training=training, step=0)
        assert err(got, want) <= tol, (name, training)
    assert err(got, base) > 0.2                                                 # ... and the mask does change the result
    left = np.ones((B, T), np.int32)
    left[1, :5] = 0                                                             # batch row 1: positions 0..4 are padding
    got, _ = m(x, attention_mask=left)
    want, _, _ = orc.forward(x, attention_mask=left)
    keep = left.astype(bool)
    assert err(got[keep], want[keep]) <= tol
    assert err(got, want) <= max(tol, 5e-3)
    # `past`: the mask covers the past keys and the new one
    got, pres = m(x, attention_mask=holes)
    want, opast, _ = orc.forward(x, attention_mask=holes)
    more = np.concatenate([holes, np.ones((B, 1), np.int32)], 1)
    got, _ = m(np.concatenate([x, x[:, :1]], 1), past=pres, attention_mask=more)
    want, _, _ = orc.forward(x[:, :1], past=opast, attention_mask=more)
    assert err(got, want) <= tol
    again, _ = m(x)
    assert np.array_equal(again, base)
    with pytest.raises(ValueError):
        m(x, attention_mask=np.ones((B, T + 1), np.int32))
    m.close()


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-4), ("bf16", 6e-2)])
def test_call_training_true_applies_the_train_step_dropout(dtype, tol):
    """self(x, training=True) (transformer.py:916-917): logits with dropout on, masks = the shared counter hash at the current
    optimizer iteration, against the oracle's forward(training=True)."""
    V, E, H, L, W, T, B = 390, 64, 4, 2, 48, 40, 2
    params = {k: v.astype(np.float32) for k, v in O.init_params(V, E, W, L, seed=21, stddev=0.1).items()}
    x, _ = O.synthetic_batch(np.random.default_rng(8), V, B, T)
    ocfg = O.Config(V, E, W, L, H, attention_dropout_rate=0.1, residual_dropout_rate=0.1)
    orc = O.OracleTransformer(ocfg, params, seed=99)
    want, _, _ = orc.forward(x, training=True, step=0)
    m = make_model((V, E, H, L, W, T, B), params, dtype, p_attn=0.1, p_resid=0.1, seed=99)
    got, _ = m(x, training=True)
    assert np.abs(got - want).max() <= tol * max(1.0, np.abs(want).max())
    off, _ = m(x)
    assert np.abs(off - want).max() > 10 * tol          # and it differs from the inference pass
    m.close()


def test_call_training_true_with_past_matches_oracle():
    """Transformer.call(inputs, past=presents, training=True) (transformer.py:696-833 allows the combination): dropout on the new
    token's embedding, on its row of the attention probabilities over all past+1 positions, and on both residual branches."""
    V, E, H, L, W, T, B = 390, 64, 4, 2, 48, 20, 2
    params = {k: v.astype(np.float32) for k, v in O.init_params(V, E, W, L, seed=23, stddev=0.1).items()}
    x, _ = O.synthetic_batch(np.random.default_rng(9), V, B, T)
    ocfg = O.Config(V, E, W, L, H, attention_dropout_rate=0.2, residual_dropout_rate=0.1)
    orc = O.OracleTransformer(ocfg, params, seed=77)
    m = make_model((V, E, H, L, W, T, B), params, "fp32", p_attn=0.2, p_resid=0.1, seed=77)
    _, opast, _ = orc.forward(x[:, :9])
    for t in range(9, 14):
        past = [np.array(p, dtype=np.float32) for p in opast]
        lg, pres = m(x[:, :t + 1], past=past, training=True)       # only the last token is used (:735-737)
        olg, _, _ = orc.forward(x[:, t:t + 1], past=opast, training=True, step=0)
        assert lg.shape == (B, 1, V) and pres[0].shape == (2, B, H, t + 1, E // H)
        assert np.abs(lg - olg).max() <= 1e-4 * max(1.0, np.abs(olg).max()), t
        off, _ = m(x[:, :t + 1], past=past)
        assert np.abs(off - olg).max() > 1e-3                      # and it differs from the dropout-free call on the same past
        _, opast, _ = orc.forward(x[:, t:t + 1], past=opast)        # next position: dropout-free presents
    m.close()


def test_ids_out_of_range_are_rejected_everywhere():
    """Every entry point refuses token ids outside [0, V): host paths before anything is launched, the device-pointer train
    step by a device-side clamp + count that cmp_train_metrics reports (evaluate / loss_and_grads / step_device)."""
    import torch
    from composer_amd import _lib
    g, cfg, params = load_golden("gA")
    V, E, H, L, W, T, B = cfg
    m = make_model(cfg, params, "fp32")
    x, y = g["x"][0].copy(), g["y"][0].copy()
    bad = x.copy(); bad[0, 3] = V
    neg = y.copy(); neg[1, 0] = -1
    for fn in (lambda: m(bad), lambda: m.loss_and_grads(bad, y), lambda: m.loss_and_grads(x, neg), lambda: m.evaluate([(x, neg)]),
               lambda: m.train_step(bad, y), lambda: m.train_step_async(x, neg)):
        with pytest.raises(ValueError):
            fn()
    xd = torch.from_numpy(bad.astype(np.int32)).cuda(); yd = torch.from_numpy(y.astype(np.int32)).cuda()
    m.train_step_device(xd.data_ptr(), yd.data_ptr(), B, T, 1e-3)
    with pytest.raises(_lib.HipLibraryError, match="outside"):
        m.last_metrics()
    m.train_step_device(torch.from_numpy(x.astype(np.int32)).cuda().data_ptr(), yd.data_ptr(), B, T, 1e-3)
    assert np.isfinite(m.last_metrics()[0])
    m.close()


def test_pipelined_train_loop_logs_the_synchronous_values(tmp_path):
    """Transformer.train submits step s+1 before it reads step s (cmp_train_step_async / cmp_train_metrics_wait): the logged
    (step, loss, accuracy) triples and the saved checkpoints equal those of one synchronous train_step per batch."""
    g, cfg, params = load_golden("gA")
    ds = [(g["x"][i], g["y"][i]) for i in range(6)]
    m = make_model(cfg, params, "fp32")
    hist = m.train(ds, (cfg[6], cfg[5]), tmp_path / "run", epochs=2, learning_rate=float(g["lr"]),
                   save_frequency_mode="global_step", save_frequency=4, max_checkpoints=3, show_progress_bar=False)
    ref = make_model(cfg, params, "fp32")
    want = [ref.train_step(x, y, float(g["lr"])) for x, y in ds]
    assert [h[0] for h in hist] == [1, 2, 3, 4, 5, 6]
    assert np.allclose([h[1] for h in hist], [w[0] for w in want], rtol=1e-5)
    assert np.allclose([h[1] for h in hist], g["losses"][:6], rtol=1e-4)
    sd, meta = __import__("composer_amd.checkpoint", fromlist=["load"]).load(str(tmp_path / "run" / "ckpt-1"))
    assert int(meta["step"]) == 4 and int(sd["optimizer/iter"]) == 4          # saved after step 4, before step 5 ran
    ref2 = make_model(cfg, params, "fp32")
    for x, y in ds[:4]:
        ref2.train_step(x, y, float(g["lr"]))
    for n in ref2.parameter_names:
        assert np.allclose(sd["model/" + n], ref2.get_parameter(n), atol=1e-6), n
    m.close(); ref.close(); ref2.close()


def test_bf16_embedding_size_not_a_multiple_of_32():
    """E = 48 (H = 3, D = 16) in bf16 mode -- the transposed weight copy has partial 32x32 tiles."""
    V, E, H, L, W, T, B = 390, 48, 3, 2, 32, 24, 2
    params = {k: v.astype(np.float32) for k, v in O.init_params(V, E, W, L, seed=13, stddev=0.1).items()}
    x, y = O.synthetic_batch(np.random.default_rng(3), V, B, T)
    orc = O.OracleTransformer(O.Config(V, E, W, L, H), params)
    loss, acc, G, logits = orc.loss_and_grads(x, y, training=False)
    m = make_model((V, E, H, L, W, T, B), params, "bf16")
    got, _ = m(x)
    assert np.abs(got - logits).max() <= 4e-2 * np.abs(logits).max()
    l2, _ = m.loss_and_grads(x, y)
    assert abs(l2 - loss) <= 2e-2 * loss
    m.close()


def test_presents_match_the_oracle():
    """Transformer.call returns (logits, presents) (transformer.py:797-806, 820-821): presents[l] = stack([key, value]) of
    layer l, [2,B,H,T,D], exported from the device through cmp_present_get."""
    g, cfg, params = load_golden("gA")
    V, E, H, L, W, T, B = cfg
    m = make_model(cfg, params, "fp32")
    x = g["x"][0]
    logits, presents = m(x)
    orc = O.OracleTransformer(O.Config(V, E, W, L, H), {k: v.astype(np.float64) for k, v in params.items()})
    ologits, opresents, _ = orc.forward(x)
    assert np.abs(logits - ologits).max() <= 1e-4
    assert len(presents) == L
    for l, p in enumerate(presents):
        assert p.shape == (2, B, H, T, E // H)
        assert np.abs(p - opresents[l]).max() <= 2e-5 * max(1.0, np.abs(opresents[l]).max()), l
    with pytest.raises(IndexError):
        presents[L]
    m.close()


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_deterministic_mode_is_bitwise_reproducible(monkeypatch, p):
    """COMPOSER_DETERMINISTIC=1 (bf16 mode): no float atomics anywhere in the step -- split-K wgrads fold per-split slabs in a
    fixed order, bias-gradient column sums and LayerNorm parameter partials are folded by one thread per column, the
    embedding scatter-add is a segmented gather.  Two runs give bitwise identical gradients for EVERY parameter and
    bitwise identical parameters after three optimizer steps; the values still match the golden loss."""
    monkeypatch.setenv("COMPOSER_DETERMINISTIC", "1")
    g, cfg, params = load_golden("gB")
    from composer_amd import _lib
    grads, finals = [], []
    for _ in range(2):
        m = make_model(cfg, params, "bf16", p_attn=p, p_resid=p, seed=7)
        loss, _ = m.loss_and_grads(g["x"][0], g["y"][0])
        grads.append({n: m.get_parameter(n, _lib.KIND_GRAD) for n in m.parameter_names})
        if p == 0.0:
            assert abs(loss - g["losses"][0]) <= 2e-2 * g["losses"][0]
        for s in range(3):
            m.train_step(g["x"][s], g["y"][s], float(g["lr"]))
        finals.append(m.get_weights())
        m.close()
    for n in grads[0]:
        assert np.array_equal(grads[0][n], grads[1][n]), n
        assert np.array_equal(finals[0][n], finals[1][n]), n
    # and the deterministic kernels compute the same gradients as the default (atomic) ones
    monkeypatch.setenv("COMPOSER_DETERMINISTIC", "0")
    m = make_model(cfg, params, "bf16", p_attn=p, p_resid=p, seed=7)
    m.loss_and_grads(g["x"][0], g["y"][0])
    for n in grads[0]:
        ref = m.get_parameter(n, _lib.KIND_GRAD)
        assert np.abs(grads[0][n] - ref).max() <= 2e-3 * np.abs(ref).max() + 1e-7, n
    m.close()


def test_checkpoint_roundtrip_resumes_bit_identically(tmp_path):
    g, cfg, params = load_golden("gA")
    m = make_model(cfg, params, "fp32")
    ds = [(g["x"][i], g["y"][i]) for i in range(4)]
    m.train(ds, (cfg[6], cfg[5]), tmp_path / "run", epochs=2, learning_rate=float(g["lr"]),
            save_frequency_mode="global_step", save_frequency=2, max_checkpoints=3, show_progress_bar=False)
    assert m.iterations == 4                       # `-e 2` runs ONE epoch (epoch counter starts at 1, transformer.py:907)
    w_ref = m.get_weights()
    nxt = m.train_step(g["x"][4], g["y"][4], float(g["lr"]))
    m2 = make_model(cfg, params, "fp32")
    m2.load_state_dict(__import__("composer_amd.checkpoint", fromlist=["load"]).load(str(tmp_path / "run" / "ckpt-2"))[0])
    for n in w_ref:
        assert np.array_equal(m2.get_parameter(n), w_ref[n]), n
    assert m2.iterations == 4
    nxt2 = m2.train_step(g["x"][4], g["y"][4], float(g["lr"]))
    assert abs(nxt[0] - nxt2[0]) < 1e-6
    m.close(); m2.close()


# ----------------------------------------------------------------------------- full-size configurations (BASELINE configs 2 and 4)
def _full_size(E, H, L, T, B, dtype, p=0.0, seed=0):
    from composer_amd.transformer import Transformer
    m = Transformer(390, E, T, L, H, attention_dropout_rate=p, residual_dropout_rate=p, dtype=dtype, seed=seed, max_batch=B, max_seq=T)
    params = {k: v.astype(np.float32) for k, v in O.init_params(390, E, T, L, seed=0).items()}
    m.set_weights(params)
    return m, params


def test_full_size_c2_properties():
    """6L/8H/d512, T=1024 (BASELINE config 2) in bf16: size-independent properties -- finite loss near ln(390) at
    init, loss falls when the same batch is repeated, eval(batch) is permutation-equivariant over batch rows."""
    V, E, H, L, W, T, B = 390, 512, 8, 6, 1024, 1024, 4
    m, params = _full_size(E, H, L, T, B, "bf16")
    rng = np.random.default_rng(1234)
    x, y = O.synthetic_batch(rng, V, B, T)
    l0, _ = m.evaluate([(x, y)])
    assert 5.9 < l0 < 7.2          # tied embeddings favour the CURRENT token, so the initial loss sits above ln(390)
    perm = np.array([2, 0, 3, 1])
    lp, _ = m.evaluate([(x[perm], y[perm])])
    assert abs(lp - l0) < 1e-3
    losses = [m.train_step(x, y, 1e-3)[0] for _ in range(6)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0] - 0.05, losses
    m.close()


def test_full_size_c2_full_length_row_matches_the_oracle():
    """BASELINE config 2 at FULL length: one T=1024 row (8 query blocks per head, the balanced q-block pairing and the 64-key
    softmax step all active) against the float64 oracle -- fp32 mode: every logit <= 1e-4; bf16 mode: loss <= 1e-2 relative and
    every parameter gradient <= 8e-2 of its largest element (the tolerance of the small-shape dropout test)."""
    from composer_amd import _lib
    V, E, H, L, W, T = 390, 512, 8, 6, 1024, 1024
    x, y = O.synthetic_batch(np.random.default_rng(77), V, 1, T)
    m32, params = _full_size(E, H, L, T, 1, "fp32")
    orc = O.OracleTransformer(O.Config(V, E, W, L, H), params)
    loss, acc, G, logits = orc.loss_and_grads(x, y, training=False)
    got, _ = m32(x)
    assert np.abs(got - logits).max() <= 1e-4, np.abs(got - logits).max()
    l32, a32 = m32.loss_and_grads(x, y)
    assert abs(l32 - loss) <= 1e-5 * loss and abs(a32 - acc) < 1e-6
    for n in m32.parameter_names:
        gr = m32.get_parameter(n, _lib.KIND_GRAD).astype(np.float64)
        assert np.abs(gr - G[n]).max() <= 5e-4 * np.abs(G[n]).max() + 1e-10, n
    m32.close()
    mb, _ = _full_size(E, H, L, T, 1, "bf16")
    lb, _ = mb.loss_and_grads(x, y)
    assert abs(lb - loss) <= 1e-2 * loss, (lb, loss)
    worst = max(np.abs(mb.get_parameter(n, _lib.KIND_GRAD).astype(np.float64) - G[n]).max() / (np.abs(G[n]).max() + 1e-12)
                for n in mb.parameter_names)
    assert worst <= 8e-2, worst
    mb.close()


def test_full_size_c2_dropout_step_matches_the_oracle():
    """The benchmark's configuration (dropout 0.1) on a full-length row: bf16 training-mode loss and gradients against the
    oracle drawing the same counter-hash masks."""
    from composer_amd import _lib
    V, E, H, L, W, T = 390, 512, 8, 6, 1024, 1024
    x, y = O.synthetic_batch(np.random.default_rng(78), V, 1, T)
    mb, params = _full_size(E, H, L, T, 1, "bf16", p=0.1, seed=5)
    orc = O.OracleTransformer(O.Config(V, E, W, L, H, attention_dropout_rate=0.1, residual_dropout_rate=0.1), params, seed=5)
    loss, acc, G, _ = orc.loss_and_grads(x, y, training=True, step=0)
    lb, _ = mb.loss_and_grads(x, y)
    assert abs(lb - loss) <= 1e-2 * loss, (lb, loss)
    worst = max(np.abs(mb.get_parameter(n, _lib.KIND_GRAD).astype(np.float64) - G[n]).max() / (np.abs(G[n]).max() + 1e-12)
                for n in mb.parameter_names)
    assert worst <= 8e-2, worst
    mb.close()


def test_full_size_c4_train_and_eval_properties():
    """BASELINE config 4: 12L/12H/d768, window 2048, bf16.  Four train steps on one batch (finite, loss falls), evaluation is
    permutation-equivariant over batch rows, and the parameter count is the surveyed one."""
    V, E, H, L, T, B = 390, 768, 12, 12, 2048, 2
    m, _ = _full_size(E, H, L, T, B, "bf16")
    assert sum(int(np.prod(m.parameter_shape(n))) for n in m.parameter_names) == 86928384
    x, y = O.synthetic_batch(np.random.default_rng(4), V, B, T)
    l0, _ = m.evaluate([(x, y)])
    assert 5.9 < l0 < 7.5
    lp, _ = m.evaluate([(x[::-1], y[::-1])])
    assert abs(lp - l0) < 1e-3
    losses = [m.train_step(x, y, 1e-3)[0] for _ in range(4)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0] - 0.05, losses
    m.close()


def test_full_size_c4_matches_the_oracle():
    """BASELINE config 4 against the oracle: bf16 loss of one FULL-LENGTH row (T=2048) <= 1e-2 relative; fp32-mode logits at
    full width and depth on T=512 <= 1e-4 absolute."""
    V, E, H, L, T = 390, 768, 12, 12, 2048
    x, y = O.synthetic_batch(np.random.default_rng(41), V, 1, T)
    mb, params = _full_size(E, H, L, T, 1, "bf16")
    orc = O.OracleTransformer(O.Config(V, E, T, L, H), params, dtype=np.float32)    # float32: [1,12,2048,2048] scores per layer
    ref, _ = orc.loss_acc(orc.forward(x)[0].astype(np.float64), y)
    got, _ = mb.evaluate([(x, y)])
    assert abs(got - ref) <= 1e-2 * ref, (got, ref)
    mb.close()
    m32, _ = _full_size(E, H, L, T, 1, "fp32")
    orc64 = O.OracleTransformer(O.Config(V, E, T, L, H), params)
    want, _, _ = orc64.forward(x[:, :512])
    lg, _ = m32(x[:, :512])
    assert np.abs(lg - want).max() <= 1e-4, np.abs(lg - want).max()
    m32.close()


def test_full_size_c5_decode():
    """BASELINE config 5: `generate` on the 6L/8H/d512 model with window 2048, KV cache + hipGraph per-token step.
    (a) greedy: the first 48 ids from a 10-id prompt equal the oracle's model(x, past=presents) loop wherever the oracle's own
    top-2 logit margin is not a near-tie (random-init logits are small; both sides carry ~1e-6 of fp32 noise);
    (b) temperature 1.0, length 1024 (the benchmark's call): ids in range, the same seed reproduces them, another seed does not,
    and the empirical distribution is not degenerate."""
    from composer_amd.transformer import Transformer
    V, E, H, L, W = 390, 512, 8, 6, 2048
    params = {k: v.astype(np.float32) for k, v in O.init_params(V, E, W, L, seed=3, stddev=0.06).items()}
    m = Transformer(V, E, W, L, H, attention_dropout_rate=0.0, residual_dropout_rate=0.0, dtype="fp32", seed=0, max_batch=1, max_seq=64)
    m.set_weights(params)
    prompt = np.random.default_rng(0).integers(0, V, 10)
    got = m.generate(prompt, 48, temperature=0.0, mode="kv").tolist()
    orc = O.OracleTransformer(O.Config(V, E, W, L, H), params)
    x = np.asarray(prompt)[None]
    logits, past, _ = orc.forward(x)
    agree, decided = 0, 0
    for i in range(48):
        z = logits[0, -1]
        top2 = np.sort(z)[-2:]
        want = int(np.argmax(z))
        if top2[1] - top2[0] > 1e-4:
            decided += 1
            agree += int(got[i] == want)
            assert got[i] == want, (i, got[i], want, top2)
        # follow the HIP ids so that one near-tie cannot derail the comparison of the later positions
        logits, past, _ = orc.forward(np.array([[got[i]]]), past=past)
    assert decided >= 40 and agree == decided
    a = m.generate(prompt, 1024, temperature=1.0, mode="kv", seed=1)
    b = m.generate(prompt, 1024, temperature=1.0, mode="kv", seed=1)
    c = m.generate(prompt, 1024, temperature=1.0, mode="kv", seed=2)
    assert len(a) == 1024 and ((a >= 0) & (a < V)).all()
    assert a.tolist() == b.tolist() and a.tolist() != c.tolist()
    assert len(set(a.tolist())) > 50
    m.close()


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_training_learns_a_deterministic_sequence(dtype):
    """End-to-end training dynamics (forward, backward, Keras Adam, dropout on): sequences in which the next id is a fixed function
    of the current one are learnt -- the loss falls from ~ln(390) to well under 0.5 and next-token accuracy passes 95 % within 300
    steps, in the parity mode and in the throughput mode alike, and the learnt model's greedy decode continues the sequence."""
    from composer_amd.transformer import Transformer
    V, E, H, L, T, B = 390, 128, 4, 2, 128, 16
    m = Transformer(V, E, T, L, H, attention_dropout_rate=0.1, residual_dropout_rate=0.1, dtype=dtype, seed=3, max_batch=B, max_seq=T)
    rng = np.random.default_rng(0)
    nxt = lambda a: (a * 7 + 3) % V

    def batch():
        s = np.empty((B, T + 1), np.int64)
        s[:, 0] = rng.integers(0, V, B)
        for t in range(T):
            s[:, t + 1] = nxt(s[:, t])
        return s[:, :-1].astype(np.int32), s[:, 1:].astype(np.int32)
    first = last = None
    for step in range(300):
        x, y = batch()
        loss, acc = m.train_step(x, y, 2e-3)
        assert np.isfinite(loss), step
        if step == 0:
            first = loss
        last = (loss, acc)
    assert first > 5.0 and last[0] < 0.5 and last[1] > 0.95, (first, last)
    x, y = batch()
    ev_loss, ev_acc = m.evaluate([(x, y)])
    assert ev_loss < 0.3 and ev_acc > 0.97, (ev_loss, ev_acc)           # dropout off at evaluation
    start = int(rng.integers(0, V))
    want = [start]
    for _ in range(20):
        want.append(int(nxt(want[-1])))
    got = m.generate(want[:4], 17, temperature=0.0, mode="kv").tolist()
    assert got == want[4:], (got, want[4:])
    m.close()
