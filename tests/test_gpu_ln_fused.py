"""-m gpu: the LayerNorm-fused block path (round 5).

For large bf16 batches no LayerNorm kernel runs inside the decoder blocks' forward pass (DecoderBlock.call,
transformer.py:574-597; LayerNormalization layers :551,563): the row statistics travel as partials (mean, M2 per 256-column
segment) written by the producing epilogue, ln_1 / ln_2 are folded into the c_attn / c_fc GEMMs
(LN(x).W + b = rstd*(x.(gamma o W)) - rstd*mean*colsum(gamma o W) + (beta.W + b)), the attention c_proj epilogue rebuilds
ln_1(x) for its residual operand, and the LayerNorm backward kernels write u = ln_1(x) / n = ln_2(r) for the weight gradients.

Every new kernel form against float64 (the oracle's formulas: O.layernorm_fwd / O.gelu / O.dropout_keep_rows), then the model:
the fused path against the unfused one and against the bf16-rounding oracle at a batch both paths can run.
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import transformer_oracle as O

pytestmark = pytest.mark.gpu
FP32, BF16 = 0, 1
V = 390
EPS = 1e-5


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


_ALIVE = []          # device tensors handed to the library as raw pointers stay referenced until the test is over


@pytest.fixture(autouse=True)
def _release():
    yield
    torch.cuda.synchronize()
    _ALIVE.clear()


def bf(a):
    _ALIVE.append(torch.as_tensor(np.asarray(a, dtype=np.float32)).to(torch.bfloat16).cuda().contiguous())
    return _ALIVE[-1]


def f32(a):
    _ALIVE.append(torch.as_tensor(np.asarray(a, dtype=np.float32)).cuda().contiguous())
    return _ALIVE[-1]


def parts_of(x64):
    """(mean, M2) of every 256-column segment of every row: float64 [rows, E/256, 2]."""
    rows, E = x64.shape
    seg = x64.reshape(rows, E // 256, 256)
    mu = seg.mean(-1)
    return np.stack([mu, ((seg - mu[..., None]) ** 2).sum(-1)], -1)


def ln64(x64, gamma, beta):
    mu = x64.mean(-1, keepdims=True)
    var = ((x64 - mu) ** 2).mean(-1, keepdims=True)
    return (x64 - mu) / np.sqrt(var + EPS) * gamma + beta


def rel(a, b):
    return np.abs(np.asarray(a, np.float64) - b).max() / (np.abs(b).max() + 1e-30)


# --------------------------------------------------------------------------------------------- embedding + partials
@pytest.mark.parametrize("E,p", [(512, 0.0), (768, 0.1), (1024, 0.1), (256, 0.0)])
def test_embedding_forward_leaves_the_rows_partial_statistics(lib, E, p):
    B, T, W = 3, 40, 64
    rng = np.random.default_rng(E)
    ids = torch.as_tensor(rng.integers(0, V, (B, T)).astype(np.int32)).cuda()
    wte, wpe = f32(rng.normal(0, 1.0, (V, E)) + 3.0), f32(rng.normal(0, 1.0, (W, E)))
    ref = torch.zeros(B * T, E, dtype=torch.bfloat16, device="cuda")
    out = torch.zeros_like(ref)
    part = torch.zeros(B * T, E // 256, 2, device="cuda")
    ck(lib, lib.cmp_k_embed_fwd(stream(), P(ids), P(wte), P(wpe), P(ref), B, T, E, 5, BF16, p, 11, 3))
    ck(lib, lib.cmp_k_embed_fwd_stats(stream(), P(ids), P(wte), P(wpe), P(out), P(part), B, T, E, 5, p, 11, 3))
    torch.cuda.synchronize()
    assert torch.equal(out, ref)                                   # the same rows, bit for bit (same mask, same rounding)
    want = parts_of(out.double().cpu().numpy())
    got = part.double().cpu().numpy()
    assert np.abs(got[..., 0] - want[..., 0]).max() < 1e-5 * (1 + np.abs(want[..., 0]).max())
    assert rel(got[..., 1], want[..., 1]) < 1e-5


# --------------------------------------------------------------------------------------------- weight side of the fold
@pytest.mark.parametrize("E,N", [(512, 1536), (768, 3072), (1024, 1024), (64, 96)])
def test_fold_prep_scales_transposes_and_sums(lib, E, N):
    rng = np.random.default_rng(E + N)
    W = rng.normal(0, 0.05, (E, N)).astype(np.float32)
    b = rng.normal(0, 0.1, N).astype(np.float32)
    g = (1 + 0.3 * rng.normal(size=E)).astype(np.float32)
    be = rng.normal(0, 0.2, E).astype(np.float32)
    WT = torch.zeros(N, E, dtype=torch.bfloat16, device="cuda")
    cs, bo = torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")
    ck(lib, lib.cmp_k_ln_fold_prep(stream(), P(f32(W)), P(f32(b)), P(f32(g)), P(f32(be)), P(WT), P(cs), P(bo), E, N))
    torch.cuda.synchronize()
    want = torch.as_tensor(g[:, None] * W).to(torch.bfloat16).T.contiguous()         # fp32 product, one rounding
    assert torch.equal(WT.cpu(), want)
    assert rel(cs.cpu().numpy(), want.double().numpy().sum(1)) < 2e-6
    assert rel(bo.cpu().numpy(), b.astype(np.float64) + be.astype(np.float64) @ W.astype(np.float64)) < 2e-6


# --------------------------------------------------------------------------------------------- GEMM epilogues
def _gemm_ln(lib, A, WT, M, N, K, bias, act, aux, resid, p, ln, out_dtype=torch.bfloat16):
    Cm = torch.zeros(M, N, dtype=out_dtype, device="cuda")
    ck(lib, lib.cmp_gemm_ln_next(*ln))
    ck(lib, lib.cmp_k_gemm(stream(), BF16, 0, 1, M, N, K, P(A), K, P(WT), K, P(Cm), N, P(bias), act, P(aux), N if aux is not None else 0,
                           P(resid), N if resid is not None else 0, 0, 1, p, 21, 9, 8))
    torch.cuda.synchronize()
    return Cm


@pytest.mark.parametrize("E", [512, 768])
@pytest.mark.parametrize("offset", [0.0, 24.0])          # rows with |mean| >> sigma: the fold subtracts two large terms
@pytest.mark.parametrize("act", [0, 1])
def test_fold_epilogue_is_layernorm_then_conv1d(lib, E, offset, act):
    """c_attn / c_fc of the fused path: the GEMM runs on the RAW rows and the gamma-scaled weight; the epilogue applies the row
    statistics.  Reference: float64 LayerNorm of the same bf16 rows, times the fp32 weight, (+ GELU, pre-activation stored)."""
    M, N = 512, 768
    rng = np.random.default_rng(E + int(offset) + act)
    x = bf(rng.normal(offset, 1.0, (M, E)) * (1 + rng.random((M, 1))))
    W = rng.normal(0, 0.05, (E, N)).astype(np.float32)
    b = rng.normal(0, 0.1, N).astype(np.float32)
    g = (1 + 0.3 * rng.normal(size=E)).astype(np.float32)
    be = rng.normal(0, 0.2, E).astype(np.float32)
    WT = torch.zeros(N, E, dtype=torch.bfloat16, device="cuda")
    cs, bo = torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")
    ck(lib, lib.cmp_k_ln_fold_prep(stream(), P(f32(W)), P(f32(b)), P(f32(g)), P(f32(be)), P(WT), P(cs), P(bo), E, N))
    x64 = x.double().cpu().numpy()
    part = f32(parts_of(x64))
    aux = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda") if act else None
    out = _gemm_ln(lib, x, WT, M, N, E, bo, act, aux, None, 0.0, (P(part), E // 256, EPS, P(cs), None, None, None))
    pre = ln64(x64, g.astype(np.float64), be.astype(np.float64)) @ W.astype(np.float64) + b
    # tight: the epilogue's own formula in float64 on the ROUNDED scaled weight (what remains is fp32 accumulation + the bf16 store)
    mu, var = x64.mean(-1, keepdims=True), x64.var(-1, keepdims=True)
    rs = 1 / np.sqrt(var + EPS)
    wt64 = WT.double().cpu().numpy()
    exact = rs * (x64 @ wt64.T) - rs * mu * wt64.sum(1) + bo.double().cpu().numpy()
    if act:
        assert rel(aux.double().cpu().numpy(), exact) < 6e-3
        assert rel(aux.double().cpu().numpy(), pre) < 1.2e-2
        exact, pre = O.gelu(exact), O.gelu(pre)
    assert rel(out.double().cpu().numpy(), exact) < 6e-3
    assert rel(out.double().cpu().numpy(), pre) < 1.2e-2               # ... and it IS LayerNorm + Conv1D within the bf16 weight rounding


@pytest.mark.parametrize("E", [512, 768])
@pytest.mark.parametrize("rebuild", [False, True])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_residual_epilogue_emits_partials_and_rebuilds_ln1(lib, E, rebuild, p):
    """Both c_proj epilogues: C = dropout(A.W + b) + residual, the residual being a stored tensor (mlp c_proj) or ln_1 of the raw
    block input rebuilt from its partials (attention c_proj); the partial statistics of C's rows go out."""
    M, K = 768, 512
    rng = np.random.default_rng(E + rebuild + int(10 * p))
    A = bf(rng.normal(0, 1.0, (M, K)))
    W = bf(rng.normal(0, 0.05, (E, K)))                              # stored [N, K]
    b = f32(rng.normal(0, 0.1, E))
    r = bf(rng.normal(2.0, 1.5, (M, E)))
    g = (1 + 0.3 * rng.normal(size=E)).astype(np.float32)
    be = rng.normal(0, 0.2, E).astype(np.float32)
    r64 = r.double().cpu().numpy()
    pin = f32(parts_of(r64))
    pout = torch.full((M, E // 256, 2), -7.0, device="cuda")
    ln = (P(pin), E // 256, EPS, None, P(f32(g)), P(f32(be)), P(pout)) if rebuild else (None, 0, 0.0, None, None, None, P(pout))
    out = _gemm_ln(lib, A, W, M, E, K, b, 0, None, r, p, ln)
    acc = A.double().cpu().numpy() @ W.double().cpu().numpy().T + b.double().cpu().numpy()
    if p > 0:
        keep = O.dropout_keep_rows(21, 9, M, E, p)
        acc = np.where(keep, acc / (1 - p), 0.0)
    res = ln64(r64, g.astype(np.float64), be.astype(np.float64)) if rebuild else r64
    if rebuild:
        res = torch.as_tensor(res).to(torch.bfloat16).double().numpy()   # the stored ln_1 output it replaces is bf16
    want = acc + res
    got = out.double().cpu().numpy()
    assert rel(got, want) < 8e-3
    sp = parts_of(got)                                                # statistics of the STORED rows
    gp = pout.double().cpu().numpy()
    assert np.abs(gp[..., 0] - sp[..., 0]).max() < 2e-5 * (1 + np.abs(sp[..., 0]).max())
    assert rel(gp[..., 1], sp[..., 1]) < 2e-5


def test_a_launch_that_cannot_carry_the_epilogue_fails_loudly(lib):
    M, N, K = 256, 256, 512
    A, W = bf(np.zeros((M, K))), bf(np.zeros((N, K)))
    part = f32(np.zeros((M, 2, 2)))
    cs = f32(np.zeros(N))
    Cm = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    for flags, n, npseg in ((4, N, 2), (2, N, 2), (8, 200, 2), (8, N, 4)):   # the 128x128 kernel, the generic kernel, a ragged N, 4 segments
        ck(lib, lib.cmp_gemm_ln_next(P(part), npseg, EPS, P(cs), None, None, None))
        rc = lib.cmp_k_gemm(stream(), BF16, 0, 1, M, n, K, P(A), K, P(W), K, P(Cm), N, None, 0, None, 0, None, 0, 0, 1, 0.0, 0, 0, flags)
        assert rc != 0 and b"LayerNorm" in lib.cmp_last_error()
    torch.cuda.synchronize()
    # ... and the request does not leak into the next call
    ck(lib, lib.cmp_k_gemm(stream(), BF16, 0, 1, M, N, K, P(A), K, P(W), K, P(Cm), N, None, 0, None, 0, None, 0, 0, 1, 0.0, 0, 0, 8))
    torch.cuda.synchronize()


# --------------------------------------------------------------------------------------------- LayerNorm backward from partials
@pytest.mark.parametrize("E", [512, 768, 1024])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_layernorm_backward_from_partials_writes_the_forward_output(lib, E, p):
    rows = 1500
    rng = np.random.default_rng(E + int(10 * p))
    x = bf(rng.normal(1.0, 2.0, (rows, E)))
    dy = bf(rng.normal(0, 1.0, (rows, E)))
    res = bf(rng.normal(0, 1.0, (rows, E)))
    g = (1 + 0.3 * rng.normal(size=E)).astype(np.float32)
    be = rng.normal(0, 0.2, E).astype(np.float32)
    x64, dy64 = x.double().cpu().numpy(), dy.double().cpu().numpy()
    part = f32(parts_of(x64))
    ws = torch.zeros(lib.cmp_k_layernorm_bwd_ws(rows, E) // 4, device="cuda")
    dx, yout, dmask = (torch.zeros(rows, E, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    dg, db, csum = torch.zeros(E, device="cuda"), torch.zeros(E, device="cuda"), torch.zeros(E, device="cuda")
    ck(lib, lib.cmp_k_layernorm_bwd_parts(stream(), P(dy), P(x), P(f32(g)), P(f32(be)), P(part), EPS, P(res), P(dx), P(yout), P(dg), P(db),
                                          P(ws), rows, E, P(dmask), P(csum), p, 5, 2))
    torch.cuda.synchronize()
    g64 = g.astype(np.float64)
    mu, var = x64.mean(-1, keepdims=True), x64.var(-1, keepdims=True)
    rs = 1 / np.sqrt(var + EPS)
    xh = (x64 - mu) * rs
    gg = dy64 * g64
    want_dx = res.double().cpu().numpy() + rs * (gg - gg.mean(-1, keepdims=True) - xh * (gg * xh).mean(-1, keepdims=True))
    assert rel(dx.double().cpu().numpy(), want_dx) < 6e-3
    assert rel(yout.double().cpu().numpy(), xh * g64 + be) < 6e-3
    assert rel(dg.cpu().numpy(), (dy64 * xh).sum(0)) < 1e-4
    assert rel(db.cpu().numpy(), dy64.sum(0)) < 1e-4
    stored = dx.double().cpu().numpy()
    keep = O.dropout_keep_rows(5, 2, rows, E, p) if p > 0 else np.ones((rows, E), bool)
    want_mask = np.where(keep, stored / (1 - p), 0.0)
    assert rel(dmask.double().cpu().numpy(), want_mask) < 5e-3          # written also with p = 0: the fused path's masked copy
    assert rel(csum.cpu().numpy(), dmask.double().cpu().numpy().sum(0)) < 1e-4


# --------------------------------------------------------------------------------------------- round 6: the backward pass without u / n
def _gelu_grad64(x):
    c, k = np.sqrt(2 / np.pi), 0.044715
    t = np.tanh(c * (x + k * x ** 3))
    return 0.5 * (1 + t) + 0.5 * x * (1 - t * t) * c * (1 + 3 * k * x * x)


@pytest.mark.parametrize("E", [512, 768])
def test_scale_epilogues_carry_the_rows_rstd_as_a_factor(lib, E):
    """The two dgrad epilogues of the raw-row backward pass (gemm.hip: LNM bit 2).  GELU' kind: C = rstd o (A.W^T * gelu'(aux)) with
    rstd merged from the LayerNorm input rows' partials, while the armed column sums stay those of the UNSCALED product (the c_fc
    bias gradient).  Residual kind: C = A.W^T + rstd o resid (du with its dr term)."""
    M, K, N = 768, 512, 1024
    rng = np.random.default_rng(E)
    A = bf(rng.normal(0, 1.0, (M, K)))
    W = bf(rng.normal(0, 0.05, (N, K)))
    aux = bf(rng.normal(0, 1.5, (M, N)))
    rows = bf(rng.normal(1.5, 2.0, (M, E)))                          # the LayerNorm input rows the statistics belong to
    r64 = rows.double().cpu().numpy()
    part = f32(parts_of(r64))
    rs = 1 / np.sqrt(r64.var(-1, keepdims=True) + EPS)
    acc = A.double().cpu().numpy() @ W.double().cpu().numpy().T
    # GELU' kind
    csum = torch.zeros(N, device="cuda")
    Cm = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    ck(lib, lib.cmp_gemm_colsum_next(P(csum)))
    ck(lib, lib.cmp_gemm_ln_scale_next(P(part), E // 256, EPS))
    ck(lib, lib.cmp_k_gemm(stream(), BF16, 0, 1, M, N, K, P(A), K, P(W), K, P(Cm), N, None, 2, P(aux), N, None, 0, 0, 1, 0.0, 0, 0, 8))
    torch.cuda.synchronize()
    prod = acc * _gelu_grad64(aux.double().cpu().numpy())
    assert rel(Cm.double().cpu().numpy(), rs * prod) < 8e-3
    assert rel(csum.cpu().numpy(), prod.sum(0)) < 2e-4
    # residual kind
    res = bf(rng.normal(0, 1.0, (M, N)))
    Cm2 = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    ck(lib, lib.cmp_gemm_ln_scale_next(P(part), E // 256, EPS))
    ck(lib, lib.cmp_k_gemm(stream(), BF16, 0, 1, M, N, K, P(A), K, P(W), K, P(Cm2), N, None, 0, None, 0, P(res), N, 0, 1, 0.0, 0, 0, 8))
    torch.cuda.synchronize()
    assert rel(Cm2.double().cpu().numpy(), acc + rs * res.double().cpu().numpy()) < 8e-3


@pytest.mark.parametrize("E", [512, 768])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_layernorm_backward_takes_a_prescaled_gradient(lib, E, p):
    """The prescaled form of the LayerNorm backward (statistics merged from the partials first): dy holds rstd o gradient (what a dgrad yields when its A operand was stored rstd-scaled); dx, the masked
    copy, its column sums and both parameter gradients are those of the plain form on the true gradient."""
    rows = 1300
    rng = np.random.default_rng(E + int(10 * p) + 1)
    x = bf(rng.normal(1.0, 2.0, (rows, E)))
    x64 = x.double().cpu().numpy()
    mu, var = x64.mean(-1, keepdims=True), x64.var(-1, keepdims=True)
    rs = 1 / np.sqrt(var + EPS)
    dy_true = rng.normal(0, 1.0, (rows, E))
    dys = bf(rs * dy_true)                                           # the stored, scaled gradient
    dy64 = dys.double().cpu().numpy() / rs                           # ... and the gradient it stands for
    res = bf(rng.normal(0, 1.0, (rows, E)))
    g = (1 + 0.3 * rng.normal(size=E)).astype(np.float32)
    be = rng.normal(0, 0.2, E).astype(np.float32)
    part = f32(parts_of(x64))
    mean_d, rstd_d = torch.zeros(rows, device="cuda"), torch.zeros(rows, device="cuda")
    ck(lib, lib.cmp_k_ln_stats_merge(stream(), P(part), E // 256, EPS, P(mean_d), P(rstd_d), rows))      # what the backward pass does for all sites at once
    torch.cuda.synchronize()
    assert np.abs(mean_d.cpu().numpy() - mu[:, 0]).max() < 1e-5 and rel(rstd_d.cpu().numpy(), rs[:, 0]) < 1e-5
    ws = torch.zeros(lib.cmp_k_layernorm_bwd_ws(rows, E) // 4, device="cuda")
    dx, dmask = (torch.zeros(rows, E, dtype=torch.bfloat16, device="cuda") for _ in range(2))
    dg, db, csum = torch.zeros(E, device="cuda"), torch.zeros(E, device="cuda"), torch.zeros(E, device="cuda")
    ck(lib, lib.cmp_k_layernorm_bwd_prescaled(stream(), P(dys), P(x), P(f32(g)), P(mean_d), P(rstd_d), P(res), P(dx), P(dg), P(db),
                                              P(ws), rows, E, P(dmask), P(csum), p, 5, 2))
    torch.cuda.synchronize()
    g64 = g.astype(np.float64)
    xh = (x64 - mu) * rs
    gg = dy64 * g64
    want_dx = res.double().cpu().numpy() + rs * (gg - gg.mean(-1, keepdims=True) - xh * (gg * xh).mean(-1, keepdims=True))
    assert rel(dx.double().cpu().numpy(), want_dx) < 6e-3
    assert rel(dg.cpu().numpy(), (dy64 * xh).sum(0)) < 2e-4
    assert rel(db.cpu().numpy(), dy64.sum(0)) < 2e-4
    keep = O.dropout_keep_rows(5, 2, rows, E, p) if p > 0 else np.ones((rows, E), bool)
    assert rel(dmask.double().cpu().numpy(), np.where(keep, dx.double().cpu().numpy() / (1 - p), 0.0)) < 5e-3


@pytest.mark.parametrize("D,H", [(64, 8), (32, 16), (128, 4)])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_attention_backward_scales_its_rows_by_the_layernorm_rstd(lib, D, H, p):
    """AttnLnRows: [dQ | dK | dV] rows stored times the rstd of their token's LayerNorm statistics, the c_attn bias gradient (column sums
    of the UNSCALED gradient) untouched -- against the plain launch on the same inputs."""
    B, T = 3, 384
    E = H * D
    rng = np.random.default_rng(D + int(10 * p))
    qkv = bf(rng.normal(0, 1.0, (B * T, 3 * E)))
    d_o = bf(rng.normal(0, 1.0, (B * T, E)))
    o = torch.zeros(B * T, E, dtype=torch.bfloat16, device="cuda")
    lse = torch.zeros(B * H * T, device="cuda")
    ck(lib, lib.cmp_k_attn_fwd(stream(), P(qkv), P(o), P(lse), B, T, H, D, 1, BF16, p, 3, 7))
    rs = rng.uniform(0.3, 3.0, (B * T, 1))
    rs_d = f32(rs[:, 0])
    outs = []
    for scaled in (False, True):
        dqkv = torch.zeros(B * T, 3 * E, dtype=torch.bfloat16, device="cuda")
        delta = torch.zeros(B * H * T, device="cuda")
        bias = torch.zeros(3 * E, device="cuda")
        ck(lib, lib.cmp_attn_bwd_bias_next(P(bias)))
        if scaled:
            ck(lib, lib.cmp_attn_bwd_ln_next(P(rs_d)))
        ck(lib, lib.cmp_k_attn_bwd(stream(), P(qkv), P(o), P(d_o), P(lse), P(delta), P(dqkv), B, T, H, D, 1, BF16, p, 3, 7))
        torch.cuda.synchronize()
        outs.append((dqkv.double().cpu().numpy(), bias.cpu().numpy().astype(np.float64)))
    (plain, bplain), (sc, bsc) = outs
    assert rel(sc, rs * plain) < 1.2e-2                              # (two bf16 roundings apart)
    assert rel(bsc, bplain) < 1e-3                                   # float atomics: not bitwise


def test_weight_gradient_on_raw_rows_becomes_the_one_on_layernorm_rows(lib):
    """wgrad_ln_fix_kernel: R = r^T . (rstd o D) accumulated on the RAW rows turns into LN(r)^T . D with nothing but R, gamma, beta and
    colsum(D): the mean term mean^T . (rstd o D) is the column mean of R, because a row's mean is the mean of that row as stored."""
    rows, E, N = 4096, 512, 768
    rng = np.random.default_rng(4)
    r64 = bf(rng.normal(0.7, 1.3, (rows, E))).double().cpu().numpy()
    Dm = rng.normal(0, 1.0, (rows, N))
    g = (1 + 0.3 * rng.normal(size=E)).astype(np.float32)
    be = rng.normal(0, 0.2, E).astype(np.float32)
    mu, var = r64.mean(-1, keepdims=True), r64.var(-1, keepdims=True)
    rs = 1 / np.sqrt(var + EPS)
    R = r64.T @ (rs * Dm)
    G = f32(R)
    ck(lib, lib.cmp_k_wgrad_ln_fix(stream(), P(G), E, N, P(f32(g)), P(f32(be)), P(f32(Dm.sum(0)))))
    torch.cuda.synchronize()
    want = ln64(r64, g.astype(np.float64), be.astype(np.float64)).T @ Dm
    assert rel(G.cpu().numpy(), want) < 2e-5


# --------------------------------------------------------------------------------------------- model level
def _model(E, H, L, T, B, p, seed=3):
    from composer_amd.transformer import Transformer
    m = Transformer(V, E, T, L, H, attention_dropout_rate=p, residual_dropout_rate=p, dtype="bf16", seed=seed, max_batch=B, max_seq=T)
    return m


def _fused(m):
    from composer_amd import _lib
    f, n = C.c_int(-1), C.c_int64(-1)
    _lib.check(_lib.load().cmp_model_path_info(m._h, C.byref(f), C.byref(n)))
    return f.value, n.value


@pytest.mark.parametrize("form", ["2", "3"])          # 2: the LayerNorm backward kernels write u / n; 3 (round 6): weight gradients on the raw rows
@pytest.mark.parametrize("E,H,L,T,B,p", [(512, 8, 2, 256, 96, 0.1), (768, 12, 2, 512, 32, 0.0)])
def test_fused_path_matches_the_unfused_path_and_the_oracle(E, H, L, T, B, p, form):
    """Same weights, same batch, same dropout masks: loss and EVERY parameter gradient of the fused path against (a) the unfused
    path of the same library (COMPOSER_LN_FUSED=0) and (b) the bf16-rounding float64 oracle on a slice of the batch small enough
    for it (rows are independent up to the 1/(B*T) loss scale); then three train steps of each path track each other."""
    from composer_amd import _lib
    params = {k: v.astype(np.float32) for k, v in O.init_params(V, E, T, L, seed=E, stddev=0.05).items()}
    rng = np.random.default_rng(E + L)
    for n in params:                                                  # LayerNorm parameters away from (1, 0): the fold must carry them
        if n.endswith("gamma"):
            params[n] = (1 + 0.2 * rng.normal(size=params[n].shape)).astype(np.float32)
        if n.endswith("beta"):
            params[n] = (0.1 * rng.normal(size=params[n].shape)).astype(np.float32)
    x, y = O.synthetic_batch(rng, V, B, T)
    res = {}
    for mode in ("1", "0"):
        os.environ["COMPOSER_LN_FUSED"] = form if mode == "1" else "0"         # 2 / 3: training passes take the fused path too
        try:
            m = _model(E, H, L, T, B, p)
            m.set_weights(params)
            loss, acc = m.loss_and_grads(x, y)
            assert _fused(m)[0] == int(mode)
            grads = {n: m.get_parameter(n, _lib.KIND_GRAD).astype(np.float64) for n in m.parameter_names}
            steps = [m.train_step(x, y, 1e-3)[0] for _ in range(3)]
            if mode == "1":
                assert _fused(m)[1] == L                               # one item table per block, built once (padding bytes in the key)
            res[mode] = (loss, acc, grads, steps)
            m.close()
        finally:
            os.environ.pop("COMPOSER_LN_FUSED", None)
    (lf, af, gf, sf), (lu, au, gu, su) = res["1"], res["0"]
    assert abs(lf - lu) <= 2e-3 * abs(lu), (lf, lu)
    bad = {n: np.abs(gf[n] - gu[n]).max() / (np.abs(gu[n]).max() + 1e-12) for n in gu}
    bad = {n: w for n, w in bad.items() if not w <= 3e-2}
    assert not bad, bad
    assert all(abs(a - b) <= 1e-2 * abs(b) for a, b in zip(sf, su)), (sf, su)
    assert sf[-1] < sf[0]


def test_fused_forward_logits_match_the_oracle_rows():
    """Inference forward (cmp_forward_logits) on the fused path against the bf16-rounding oracle, row by row for a few rows of a batch
    large enough to take it (rows of a batch are independent in the forward pass)."""
    E, H, L, T, B = 512, 8, 2, 256, 96
    params = {k: v.astype(np.float32) for k, v in O.init_params(V, E, T, L, seed=5, stddev=0.05).items()}
    rng = np.random.default_rng(9)
    for n in params:
        if n.endswith("gamma"):
            params[n] = (1 + 0.2 * rng.normal(size=params[n].shape)).astype(np.float32)
        if n.endswith("beta"):
            params[n] = (0.1 * rng.normal(size=params[n].shape)).astype(np.float32)
    x, _ = O.synthetic_batch(rng, V, B, T)
    m = _model(E, H, L, T, B, 0.0)
    m.set_weights(params)
    logits = m(x, training=False)[0]
    assert _fused(m)[0] == 1
    m.close()
    orc = O.OracleTransformer(O.Config(V, E, T, L, H), params, emulate_bf16=True)
    for b in (0, 41, 95):
        want = np.asarray(orc.forward(x[b:b + 1])[0])
        got = np.asarray(logits)[b:b + 1]
        assert np.abs(got - want).max() <= 3e-2 * np.abs(want).max(), b


def test_fused_forward_under_a_communicator_takes_items_dynamically():
    """In a data-parallel job the persistent GEMMs claim their items from per-XCD counters (RCCL kernels may hold CUs); the fold
    epilogues' hand-counted wait for their LDS-DMA image must hold on that path too: evaluation on a one-rank communicator equals the
    plain run bit for bit (same kernels, same order of every sum)."""
    from composer_amd.transformer import Transformer
    E, H, L, T, B = 512, 8, 2, 256, 96
    params = {k: v.astype(np.float32) for k, v in O.init_params(V, E, T, L, seed=11, stddev=0.05).items()}
    x, y = O.synthetic_batch(np.random.default_rng(3), V, B, T)
    out = []
    for dp in (False, True):
        m = _model(E, H, L, T, B, 0.0)
        m.set_weights(params)
        if dp:
            m.init_data_parallel(0, 1, Transformer.new_unique_id())
        logits = m(x, training=False)[0]
        assert _fused(m)[0] == 1
        out.append((np.asarray(logits).copy(), m.evaluate([(x, y)])))
        m.close()
    assert np.array_equal(out[0][0], out[1][0])
    assert out[0][1] == out[1][1]


def test_counting_the_launches_of_a_step_changes_nothing():
    """cmp_train_step_launches captures one step on the stream, counts the graph's kernel nodes and drops it: the model's
    trajectory is the one of a model that was never asked."""
    from composer_amd import _lib
    E, H, L, T, B = 256, 16, 2, 256, 2
    params = {k: v.astype(np.float32) for k, v in O.init_params(V, E, T, L, seed=2, stddev=0.05).items()}
    x, y = O.synthetic_batch(np.random.default_rng(5), V, B, T)
    xd, yd = torch.as_tensor(x).cuda(), torch.as_tensor(y).cuda()
    traj = []
    for count in (False, True):
        m = _model(E, H, L, T, B, 0.1)
        m.set_weights(params)
        losses = [m.train_step(x, y, 1e-3)[0]]
        if count:
            nk, no = C.c_int(0), C.c_int(0)
            _lib.check(_lib.load().cmp_train_step_launches(m._h, C.c_void_p(xd.data_ptr()), C.c_void_p(yd.data_ptr()), B, T, C.byref(nk), C.byref(no)))
            assert 30 < nk.value < 200 and no.value >= 1, (nk.value, no.value)
        losses += [m.train_step(x, y, 1e-3)[0] for _ in range(3)]
        traj.append(losses)
        m.close()
    assert np.allclose(traj[0], traj[1], rtol=2e-3), traj        # (bf16 + float atomics: not bitwise)


@pytest.mark.parametrize("E", [512, 768])
def test_fold_epilogue_with_fp32_output_and_a_ragged_last_column(lib, E):
    """ln_f folded into the tied-logits GEMM (transformer.py:811, 818): fp32 output, N = 390 columns in rows of 448 -- the last tile
    column is ragged and the padding columns must stay untouched."""
    M, N, ldc = 512, 390, 448
    rng = np.random.default_rng(E)
    x = bf(rng.normal(0.5, 1.5, (M, E)))
    wte = rng.normal(0, 0.05, (N, E)).astype(np.float32)
    g = (1 + 0.3 * rng.normal(size=E)).astype(np.float32)
    be = rng.normal(0, 0.2, E).astype(np.float32)
    WT = bf(g[None, :] * wte)                                         # [N, E]: the gamma-scaled copy of wte
    wt64 = WT.double().cpu().numpy()
    cs = f32(np.concatenate([wt64.sum(1), np.zeros(512 - N)]))
    bias = f32(np.concatenate([wte.astype(np.float64) @ be.astype(np.float64), np.zeros(512 - N)]))
    x64 = x.double().cpu().numpy()
    part = f32(parts_of(x64))
    out = torch.full((M, ldc), -3.0, dtype=torch.float32, device="cuda")
    ck(lib, lib.cmp_gemm_ln_next(P(part), E // 256, EPS, P(cs), None, None, None))
    ck(lib, lib.cmp_k_gemm(stream(), BF16, 0, 1, M, N, E, P(x), E, P(WT), E, P(out), ldc, P(bias), 0, None, 0, None, 0, 1, 1, 0.0, 0, 0, 8))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    want = ln64(x64, g.astype(np.float64), be.astype(np.float64)) @ wte.astype(np.float64).T
    assert rel(got[:, :N], want) < 1.2e-2
    mu, var = x64.mean(-1, keepdims=True), x64.var(-1, keepdims=True)
    rs = 1 / np.sqrt(var + EPS)
    exact = rs * (x64 @ wt64.T) - rs * mu * wt64.sum(1) + bias.double().cpu().numpy()[:N]
    assert rel(got[:, :N], exact) < 2e-5                              # fp32 output: only the accumulation order is left
    assert np.all(got[:, N:] == -3.0)


def test_hidden_states_after_a_pass_that_folded_ln_f():
    """output_hidden_states on the fused path: the ln_f output is not written by the pass (ln_f lives in the logits GEMM) and is
    produced when asked for -- it must be LayerNorm(last block output) all the same."""
    from composer_amd.transformer import Transformer
    E, H, L, T, B = 512, 8, 2, 256, 96
    params = {k: v.astype(np.float32) for k, v in O.init_params(V, E, T, L, seed=4, stddev=0.05).items()}
    rng = np.random.default_rng(1)
    params["ln_f/gamma"] = (1 + 0.2 * rng.normal(size=E)).astype(np.float32)
    params["ln_f/beta"] = (0.1 * rng.normal(size=E)).astype(np.float32)
    x, _ = O.synthetic_batch(rng, V, B, T)
    m = Transformer(V, E, T, L, H, attention_dropout_rate=0.0, residual_dropout_rate=0.0, dtype="bf16", seed=3, max_batch=B, max_seq=T,
                    output_hidden_states=True)
    m.set_weights(params)
    logits, _, hidden = m(x, training=False)
    assert _fused(m)[0] == 1
    hf = hidden[L]                                                   # hidden[i < L] are the blocks' inputs, hidden[L] the ln_f output
    m.close()
    os.environ["COMPOSER_LN_FUSED"] = "0"
    try:
        m2 = Transformer(V, E, T, L, H, attention_dropout_rate=0.0, residual_dropout_rate=0.0, dtype="bf16", seed=3, max_batch=B, max_seq=T,
                         output_hidden_states=True)
        m2.set_weights(params)
        logits2, _, hidden2 = m2(x, training=False)
        assert _fused(m2)[0] == 0
        m2.close()
    finally:
        os.environ.pop("COMPOSER_LN_FUSED", None)
    assert np.abs(hf - hidden2[L]).max() <= 4e-2 * np.abs(hidden2[L]).max()
    assert np.abs(np.asarray(logits) - np.asarray(logits2)).max() <= 3e-2 * np.abs(np.asarray(logits2)).max()


def test_fused_forward_whose_logits_launch_has_fewer_tiles_than_its_block_launches():
    """E = 768, V = 390 at 16 384 tokens -- the blocks' [M, E] GEMMs have 64 x 3 = 192 tiles of 256 x 256 (the fused
    path's threshold), the tied-logits GEMM only 64 x 2 = 128: by size it would go to the 128 x 128 kernel, which has no LayerNorm
    epilogue, and the folded ln_f made the pass fail.  The fold launch now names its kernel; the pass must run, take the fused path and
    give the unfused path's logits and loss."""
    E, H, L, T, B = 768, 12, 1, 512, 32
    params = {k: v.astype(np.float32) for k, v in O.init_params(V, E, T, L, seed=8, stddev=0.05).items()}
    rng = np.random.default_rng(12)
    params["ln_f/gamma"] = (1 + 0.2 * rng.normal(size=E)).astype(np.float32)
    params["ln_f/beta"] = (0.1 * rng.normal(size=E)).astype(np.float32)
    x, y = O.synthetic_batch(rng, V, B, T)
    out = {}
    for mode in ("1", "0"):
        if mode == "0":
            os.environ["COMPOSER_LN_FUSED"] = "0"
        try:
            m = _model(E, H, L, T, B, 0.0)
            m.set_weights(params)
            logits = np.asarray(m(x, training=False)[0]).copy()
            assert _fused(m)[0] == int(mode)
            out[mode] = (logits, m.evaluate([(x, y)]))
            m.close()
        finally:
            os.environ.pop("COMPOSER_LN_FUSED", None)
    assert np.abs(out["1"][0] - out["0"][0]).max() <= 3e-2 * np.abs(out["0"][0]).max()
    assert abs(out["1"][1][0] - out["0"][1][0]) <= 2e-3 * abs(out["0"][1][0])
