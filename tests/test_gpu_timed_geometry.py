"""-m gpu: parity AT THE LAUNCH GEOMETRY bench.py TIMES.

The benchmark runs M = B*T = 131 072 tokens (C2: 6L/8H/d512, seq 1024, B = 128 per GPU) and 65 536 (C4: 12L/12H/d768, seq 2048,
B = 32): persistent GEMMs over >= 4 rounds of 256x256 items, weight-gradient GEMMs that contract over 131 072 tokens in 16-48
splits, attention grids of 1 024 (batch, head) groups.  The other GPU tests stop at B = 4 / K = 4 096 / 64 groups; these run the
timed shapes themselves:

* model level -- one full-length row tiled B times.  The loss is a mean over B*T positions (transformer.py:888,918), so the
  batch of B identical rows has the loss and EVERY parameter gradient of the single row, which
  tests/test_gpu_model.py::test_full_size_c2_full_length_row_matches_the_oracle ties to the float64 oracle (and which is re-run
  here at B = 1 on the HIP path as the comparison value).  fp32 and bf16, dropout 0.
* cmp_k_gemm wgrad at the four per-layer shapes with K = 131 072 and the split counts model.hip::wgrad_splits picks, against a
  float64 CPU product.
* cmp_k_attn_fwd / cmp_k_attn_bwd at B*H = 1 024 groups (C2) and 384 groups of T = 2 048 (C4), dropout 0.1: five (batch, head)
  groups checked in full (o, lse, dQ, dK, dV) against a float64 reference drawing the oracle's masks.
"""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from oracle import transformer_oracle as O

pytestmark = pytest.mark.gpu
FP32, BF16 = 0, 1
V = 390


@pytest.fixture(scope="module")
def lib():
    from composer_amd import _lib
    l = _lib.load()
    _lib.require_gpu()
    return l


def ck(lib, rc):
    assert rc == 0, lib.cmp_last_error().decode()


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _model(E, H, L, T, B, dtype):
    from composer_amd.transformer import Transformer
    m = Transformer(V, E, T, L, H, attention_dropout_rate=0.0, residual_dropout_rate=0.0, dtype=dtype, seed=0, max_batch=B, max_seq=T)
    m.set_weights({k: v.astype(np.float32) for k, v in O.init_params(V, E, T, L, seed=0).items()})
    return m


def _grads(m):
    from composer_amd import _lib
    return {n: m.get_parameter(n, _lib.KIND_GRAD).astype(np.float64) for n in m.parameter_names}


@pytest.mark.parametrize("cfg,dtype,tol_loss,tol_grad", [
    ("c2", "fp32", 2e-6, 3e-4), ("c2", "bf16", 2e-4, 2e-2), ("c4", "bf16", 2e-4, 2e-2)])
def test_tiled_row_at_the_timed_batch_has_the_single_row_loss_and_gradients(cfg, dtype, tol_loss, tol_grad):
    """C2 at B = 128 (fp32 and bf16) and C4 at B = 32 (bf16): the shapes of bench.py's `value`, `b32`... `c4` lines."""
    E, H, L, T, B = {"c2": (512, 8, 6, 1024, 128), "c4": (768, 12, 12, 2048, 32)}[cfg]
    x, y = O.synthetic_batch(np.random.default_rng(77), V, 1, T)          # the row of test_full_size_c2_full_length_row_matches_the_oracle
    m1 = _model(E, H, L, T, 1, dtype)
    l1, a1 = m1.loss_and_grads(x, y)
    g1 = _grads(m1)
    m1.close()
    mb = _model(E, H, L, T, B, dtype)
    lb, ab = mb.loss_and_grads(np.tile(x, (B, 1)), np.tile(y, (B, 1)))
    gb = _grads(mb)
    assert abs(lb - l1) <= tol_loss * abs(l1), (lb, l1)
    assert abs(ab - a1) < 1e-6
    worst = {}
    for n in g1:
        worst[n] = np.abs(gb[n] - g1[n]).max() / (np.abs(g1[n]).max() + 1e-12)
    bad = {n: w for n, w in worst.items() if not w <= tol_grad}
    assert not bad, bad
    # ... and a whole train step (forward, backward, Adam) at that geometry: the loss AFTER one update equals the single-row
    # run's (elements whose gradient is rounding noise may step either way under Adam's sign-like first update; they do not
    # move the loss to first order)
    if cfg == "c2":
        m1 = _model(E, H, L, T, 1, dtype)
        s1 = [m1.train_step(x, y, 1e-3)[0] for _ in range(2)]
        sb = [mb.train_step(np.tile(x, (B, 1)), np.tile(y, (B, 1)), 1e-3)[0] for _ in range(2)]
        assert abs(sb[0] - s1[0]) <= tol_loss * abs(s1[0]) and sb[1] < sb[0], (sb, s1)
        assert abs(sb[1] - s1[1]) <= (2e-3 if dtype == "fp32" else 2e-2) * abs(s1[1]), (sb, s1)
        m1.close()
    mb.close()


@pytest.mark.parametrize("B,T,E,H,L,p", [(32, 128, 64, 2, 2, 0.15), (9, 512, 128, 2, 1, 0.0), (48, 96, 96, 4, 2, 0.1)])
def test_batches_that_take_the_grouped_and_sorted_kernels_match_the_oracle(B, T, E, H, L, p):
    """From 4 096 tokens on the backward pass changes kernels: the embedding gradient goes through the sorted form, and in bf16 the
    four weight gradients of a block through ONE grouped launch (tokens a multiple of 32).  Medium batches at which the float64
    oracle is still quick: fp32 and bf16 (against the bf16-rounding oracle) loss and EVERY parameter gradient, dropout on in
    two of the three, a head size without a kernel of its own (24) in the third."""
    from composer_amd import _lib
    from composer_amd.transformer import Transformer
    W = T
    params = {k: v.astype(np.float32) for k, v in O.init_params(V, E, W, L, seed=B + T, stddev=0.08).items()}
    rng = np.random.default_rng(B * T)
    x, y = O.synthetic_batch(rng, V, B, T)
    x[:, ::3] = x[0, 0]                                    # a third of the tokens share one id: a long run in the sorted token list
    ocfg = O.Config(V, E, W, L, H, attention_dropout_rate=p, residual_dropout_rate=p)
    for dtype in ("fp32", "bf16"):
        orc = O.OracleTransformer(ocfg, params, seed=7, emulate_bf16=(dtype == "bf16"))
        loss, acc, G, _ = orc.loss_and_grads(x, y, training=p > 0, step=0)
        m = Transformer(V, E, W, L, H, attention_dropout_rate=p, residual_dropout_rate=p, dtype=dtype, seed=7, max_batch=B, max_seq=T)
        m.set_weights(params)
        l2, _ = m.loss_and_grads(x, y)
        assert abs(l2 - loss) <= (2e-5 if dtype == "fp32" else 2e-2) * abs(loss), (dtype, l2, loss)
        worst = {n: np.abs(m.get_parameter(n, _lib.KIND_GRAD).astype(np.float64) - G[n]).max() / (np.abs(G[n]).max() + 1e-12)
                 for n in m.parameter_names}
        bad = {n: w for n, w in worst.items() if not w <= (5e-4 if dtype == "fp32" else 4e-2)}
        assert not bad, (dtype, bad)
        m.close()


@pytest.mark.parametrize("name,m,n", [("mlp c_proj", 2048, 512), ("c_fc", 512, 2048), ("attn c_proj", 512, 512), ("c_attn", 512, 1536)])
def test_wgrad_contracts_over_131072_tokens_like_a_float64_product(lib, name, m, n):
    """dW = X^T.dY with X stored [tokens, m], dY [tokens, n], K = 131 072 tokens, split-K by f32 atomics with the split count the
    model uses (model.hip::wgrad_splits: 256 CUs / 256x256 tiles, at least 2)."""
    K = 131072
    tiles = ((m + 255) // 256) * ((n + 255) // 256)
    splitk = max(2, min(max(1, 256 // tiles) if tiles >= 8 else max(1, 768 // (((m + 127) // 128) * ((n + 127) // 128))), K // 256))
    g = torch.Generator().manual_seed(m + n)
    A = torch.randn(K, m, generator=g).to(torch.bfloat16)
    Bm = torch.randn(K, n, generator=g).to(torch.bfloat16)
    ref = A.double().numpy().T @ Bm.double().numpy()
    Ad, Bd = A.cuda(), Bm.cuda()
    out = torch.zeros(m, n, device="cuda", dtype=torch.float32)
    ck(lib, lib.cmp_k_gemm(stream(), BF16, 1, 0, m, n, K, P(Ad), m, P(Bd), n, P(out), n, None, 0, None, 0, None, 0, 1, splitk, 0.0, 0, 0, 0))
    torch.cuda.synchronize()
    got = out.double().cpu().numpy()
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err <= 5e-5, (name, splitk, err)       # products of bf16 pairs are exact in fp32; only the summation order differs
    # accumulate semantics of split-K (the model adds into a pre-zeroed gradient): a second launch doubles the result
    ck(lib, lib.cmp_k_gemm(stream(), BF16, 1, 0, m, n, K, P(Ad), m, P(Bd), n, P(out), n, None, 0, None, 0, None, 0, 1, splitk, 0.0, 0, 0, 0))
    torch.cuda.synchronize()
    assert np.abs(out.double().cpu().numpy() - 2 * ref).max() / np.abs(ref).max() <= 1e-4


@pytest.mark.parametrize("B,H,T", [(128, 8, 1024), (32, 12, 2048)])
def test_attention_at_the_timed_group_count(lib, B, H, T):
    """B*H = 1 024 groups of T = 1 024 (C2) and 384 groups of T = 2 048 (C4), D = 64, bf16, dropout 0.1 (the benchmark's rate):
    first, last and three groups in between against the float64 reference with the oracle's masks."""
    D, p, dtype = 64, 0.1, BF16
    E = H * D
    g = torch.Generator().manual_seed(B + T)
    qkv = torch.randn(B * T, 3 * E, generator=g).to(torch.bfloat16).cuda()
    do = torch.randn(B * T, E, generator=g).to(torch.bfloat16).cuda()
    o = torch.zeros(B * T, E, device="cuda", dtype=torch.bfloat16)
    lse = torch.zeros(B * H * T, device="cuda")
    ck(lib, lib.cmp_k_attn_fwd(stream(), P(qkv), P(o), P(lse), B, T, H, D, 1, dtype, p, 77, 9))
    dqkv = torch.zeros(B * T, 3 * E, device="cuda", dtype=torch.bfloat16)
    delta = torch.zeros(B * H * T, device="cuda")
    ck(lib, lib.cmp_k_attn_bwd(stream(), P(qkv), P(o), P(do), P(lse), P(delta), P(dqkv), B, T, H, D, 1, dtype, p, 77, 9))
    torch.cuda.synchronize()
    tri = torch.tril(torch.ones(T, T, dtype=torch.float64))
    lh = lse.cpu().reshape(B, H, T)

    def rel(a, b):
        return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()

    for b, h in ((0, 0), (B // 3, 5), (B // 2, 2), (B - 2, H - 1), (B - 1, H - 1)):
        rows = slice(b * T, (b + 1) * T)
        x = torch.stack([qkv[rows, j * E + h * D:j * E + (h + 1) * D].double().cpu() for j in range(3)], 1).requires_grad_(True)   # [T, 3, D]
        w = (x[:, 0] @ x[:, 1].T) * (1.0 / math.sqrt(D))
        w = w * tri - 1e4 * (1 - tri)
        pr = torch.softmax(w, -1)
        keep = O.dropout_keep_rows(77, 9, T, T, p, row0=(b * H + h) * T)
        oref = (pr * torch.tensor(keep / (1 - p))) @ x[:, 2]
        oref.backward(do[rows, h * D:(h + 1) * D].double().cpu())
        assert rel(o[rows, h * D:(h + 1) * D].double().cpu(), oref.detach()) < 3.6e-2, (b, h)
        assert rel(lh[b, h].double(), torch.logsumexp(w, -1).detach()) < 2e-2, (b, h)
        for j, name in enumerate(("dq", "dk", "dv")):
            got = dqkv[rows, j * E + h * D:j * E + (h + 1) * D].double().cpu()
            assert rel(got, x.grad[:, j]) < 7.2e-2, (b, h, name)
    # the (batch, head) groups that were not compared in float64 hold finite values of the right scale
    assert torch.isfinite(o.float()).all() and torch.isfinite(dqkv.float()).all()
    assert o.float().abs().mean().item() > 1e-3 and dqkv.float().abs().mean().item() > 1e-4
