"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU restatement (numpy, float64 by default) of the reference hot path:
galacticglum/composer `composer/models/transformer.py` (Transformer teacher-forced
training step + autoregressive decode).  Only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import this module.  The shipped package
`composer_amd` never imports it and fails loudly when its HIP library is missing.

PARITY UNPINNED: the arithmetic of the reference path lives in TensorFlow
(`tensorflow-gpu`, unpinned in `environment.yml:13`; TF 2.1/2.2 era), which is neither
vendored under /root/reference nor installable offline, and the reference's own tests
(`tests/test_sequences.py`) hold no golden vector for this path.  This oracle restates
the reference's call sites op by op plus the published TF/Keras semantics of each op; it
is cross-checked against an independent torch-autograd restatement in
`tests/test_oracle.py` and frozen into `tests/golden/*.npz` by `tests/golden/make_golden.py`.

Every function cites the reference file:line it follows (paths relative to /root/reference).
"""
import math
import numpy as np

# --------------------------------------------------------------------------------------
# Parameter surface (checkpoint names / shapes) -- transformer.py:114-116,189-190,663-694
# --------------------------------------------------------------------------------------

def param_specs(V, E, W, L):
    """Ordered (name, shape, kind) list. kind in {'normal','ones','zeros'}.
    Shapes: wte [V,E] (transformer.py:114), wpe [W,E] (:675-679), Conv1D weight
    [in,out] and bias [1,out] (:189-190), LayerNormalization gamma/beta [E] (:551,563,694)."""
    s = [("wte/weight", (V, E), "normal"), ("wpe/embeddings", (W, E), "normal")]
    for i in range(L):
        p = "decoder_blocks/%d/" % i
        s += [
            (p + "ln_1/gamma", (E,), "ones"), (p + "ln_1/beta", (E,), "zeros"),
            (p + "attn/c_attn/weight", (E, 3 * E), "normal"), (p + "attn/c_attn/bias", (1, 3 * E), "zeros"),
            (p + "attn/c_proj/weight", (E, E), "normal"), (p + "attn/c_proj/bias", (1, E), "zeros"),
            (p + "ln_2/gamma", (E,), "ones"), (p + "ln_2/beta", (E,), "zeros"),
            (p + "mlp/c_fc/weight", (E, 4 * E), "normal"), (p + "mlp/c_fc/bias", (1, 4 * E), "zeros"),
            (p + "mlp/c_proj/weight", (4 * E, E), "normal"), (p + "mlp/c_proj/bias", (1, E), "zeros"),
        ]
    s += [("ln_f/gamma", (E,), "ones"), ("ln_f/beta", (E,), "zeros")]
    return s


def init_params(V, E, W, L, seed=0, mean=0.0, stddev=0.02, dtype=np.float64):
    """TruncatedNormal(mean, stddev) resampled outside +-2 sigma (Keras initializer used at
    transformer.py:115,188,670-673), gamma=1, beta=0, bias=0.  Uses numpy's own stream: TF's
    Philox stream is not reproducible outside TF, parity runs load identical weights instead."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape, kind in param_specs(V, E, W, L):
        if kind == "normal":
            a = rng.standard_normal(shape)
            bad = np.abs(a) > 2.0
            while bad.any():
                a[bad] = rng.standard_normal(int(bad.sum()))
                bad = np.abs(a) > 2.0
            out[name] = (mean + stddev * a).astype(dtype)
        elif kind == "ones":
            out[name] = np.ones(shape, dtype)
        else:
            out[name] = np.zeros(shape, dtype)
    return out


# --------------------------------------------------------------------------------------
# Stateless dropout mask shared bit-for-bit with the HIP kernels (csrc/common.h: attn_row_hash / attn_elem_hash / apply_drop).
# The reference uses tf.nn.dropout (keep-scale 1/(1-p), transformer.py:271-272,361,444,496,506,
# 681,794); TF's random stream is not reproducible, so both sides use this counter hash.
# --------------------------------------------------------------------------------------

def _mix32(x):
    x = x.astype(np.uint64) & 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def dropout_keep_rows(seed, stream, nrows, ncols, p, row0=0):
    """Mask [nrows, ncols] (csrc/common.h: attn_row_hash / attn_elem_hash; apply_drop): one full hash per row; the 4
    columns of a group share the xor base and differ by a 24-bit multiplier:
        keep[r, c] = (((rowhash(r) ^ ((c >> 2) * 0x9E3779B1)) & 0xFFFFFF) * C24[c & 3]) mod 2^32 >= p * 2^32.
    Used by every dropout site: rows are tokens (b*T + t) and columns features for the embedding / c_proj / MLP
    outputs; rows are (batch*head, query) and columns keys for the attention probabilities."""
    rows = np.arange(row0, row0 + nrows, dtype=np.uint64)        # row0: a window of a larger mask (full-size tests)
    rowh = _mix32(_mix32(rows ^ np.uint64(seed & 0xFFFFFFFF)) ^ np.uint64(stream & 0xFFFFFFFF))
    cols = np.arange(ncols, dtype=np.uint64)
    cq = np.array([0xEBCA6B, 0xB2AE35, 0xD4EB2F, 0x5667B1], dtype=np.uint64)       # 24-bit multipliers
    x = (rowh[:, None] ^ (((cols >> 2) * 0x9E3779B1) & 0xFFFFFFFF)[None, :]) & 0xFFFFFF
    x = (x * cq[cols & 3][None, :]) & 0xFFFFFFFF
    thr = np.uint64(min(int(p * 4294967296.0), 0xFFFFFFFF))
    return x >= thr


def dropout_keep_attn(seed, stream, BH, T, p, past_len=0):
    """Attention-probability mask [BH, T(query), past_len + T(key)]: the last T rows of the square mask over past_len + T
    positions (with `past`, transformer.py:423-426, the T new queries sit at positions past_len .. past_len + T - 1)."""
    Tt = past_len + T
    return dropout_keep_rows(seed, stream, BH * Tt, Tt, p).reshape(BH, Tt, Tt)[:, past_len:, :]


def dropout_stream(step, layer, site):
    """site: 0 embed, 1 attention probabilities, 2 attn c_proj output, 3 mlp output."""
    return ((step * 64 + layer) * 4 + site) & 0xFFFFFFFF


# --------------------------------------------------------------------------------------
# Primitive ops
# --------------------------------------------------------------------------------------

_GELU_C = math.sqrt(2.0 / math.pi)


def gelu(x):
    """transformer.py:35-40: 0.5*x*(1+tanh(sqrt(2/pi)*(x+0.044715*x^3)))."""
    return 0.5 * x * (1.0 + np.tanh(_GELU_C * (x + 0.044715 * x ** 3)))


def gelu_grad(x):
    t = np.tanh(_GELU_C * (x + 0.044715 * x ** 3))
    return 0.5 * (1.0 + t) + 0.5 * x * (1.0 - t * t) * _GELU_C * (1.0 + 3 * 0.044715 * x * x)


def layernorm_fwd(x, gamma, beta, eps):
    """Keras LayerNormalization non-fused path (transformer.py:551,563,694): biased variance over
    the last axis, x*inv + (beta - mean*inv), inv = rsqrt(var+eps)*gamma."""
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = (x - mu) * rstd
    return xhat * gamma + beta, (xhat, rstd)


def layernorm_bwd(dy, cache, gamma):
    xhat, rstd = cache
    E = dy.shape[-1]
    dgamma = (dy * xhat).reshape(-1, E).sum(0)
    dbeta = dy.reshape(-1, E).sum(0)
    g = dy * gamma
    dx = rstd * (g - g.mean(-1, keepdims=True) - xhat * (g * xhat).mean(-1, keepdims=True))
    return dx, dgamma, dbeta


def causal_mask(nd, ns, dtype):
    """transformer.py:290-301: m[i,j] = (i >= j - ns + nd)."""
    i = np.arange(nd)[:, None]
    j = np.arange(ns)
    return (i >= j - ns + nd).astype(dtype)


class Config:
    def __init__(self, vocab_size, embedding_size, window_size, decoder_layers_count,
                 attention_head_count, layer_normalization_epsilon=1e-5, scale=True,
                 use_layer_normalization=True, attention_dropout_rate=0.0,
                 residual_dropout_rate=0.0):
        assert embedding_size % attention_head_count == 0  # transformer.py:255
        self.V, self.E, self.W = vocab_size, embedding_size, window_size
        self.L, self.H = decoder_layers_count, attention_head_count
        self.D = embedding_size // attention_head_count
        self.eps = layer_normalization_epsilon
        self.scale = scale
        self.use_ln = use_layer_normalization
        self.p_attn = attention_dropout_rate
        self.p_resid = residual_dropout_rate


def round_bf16(a):
    """Round-to-nearest-even to bfloat16 precision (returned in the input's float type).  Used only by the
    `emulate_bf16` mode below: the checker then rounds where the HIP bf16 kernels round."""
    a = np.asarray(a)
    f = np.ascontiguousarray(a, dtype=np.float32)
    u = f.view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return u.view(np.float32).reshape(a.shape).astype(a.dtype if a.dtype.kind == "f" else np.float64)


class OracleTransformer:
    """Restatement of `Transformer` (transformer.py:599-960).

    emulate_bf16=True keeps the reference's arithmetic but rounds to bfloat16 at the points where the HIP throughput
    mode stores bf16 (activations written to HBM, the bf16 weight shadow feeding every GEMM, the
    probabilities / score gradients feeding the attention MFMAs, the bf16 gradient tensors of the backward pass);
    accumulation, LayerNorm statistics, softmax, logits and the loss stay in the oracle's float type.  It exists so
    that the bf16 kernels can be held to a tolerance that reflects summation order only, not bf16 storage itself."""

    def __init__(self, cfg, params, dtype=np.float64, seed=0, emulate_bf16=False):
        self.cfg = cfg
        self.dtype = dtype
        self.emulate_bf16 = bool(emulate_bf16)
        self.R = round_bf16 if emulate_bf16 else (lambda a: a)
        self.p = {k: np.array(v, dtype=dtype) for k, v in params.items()}
        self.m = {k: np.zeros_like(v) for k, v in self.p.items()}
        self.v = {k: np.zeros_like(v) for k, v in self.p.items()}
        self.iterations = 0          # Keras optimizer.iterations
        self.seed = seed

    def _w(self, name):
        """GEMM weight operand: the fp32 master value, or its bf16 shadow when emulating the throughput mode."""
        return self.R(self.p[name])

    # ---------------------------------------------------------------- forward
    def _dropout(self, x, p, step, layer, site, training):
        if not training or p <= 0.0:
            return x, None
        if site == 1:      # attention probabilities [B,H,T,past_len+T]
            keep = dropout_keep_attn(self.seed, dropout_stream(step, layer, site), x.shape[0] * x.shape[1],
                                     x.shape[2], p, past_len=x.shape[3] - x.shape[2]).reshape(x.shape)
        else:
            keep = dropout_keep_rows(self.seed, dropout_stream(step, layer, site), x.size // x.shape[-1],
                                     x.shape[-1], p).reshape(x.shape)
        scale = 1.0 / (1.0 - p)
        m = keep.astype(self.dtype) * scale
        return x * m, m

    def forward(self, x, past=None, training=False, step=0, keep_cache=False, position_ids=None, token_type_ids=None,
                attention_mask=None):
        """Transformer.call (transformer.py:696-833).  x int [B,T].  past: list of L arrays
        [2,B,H,Tp,D] or None.  Returns logits [B,T,V], presents (list of [2,B,H,Tk,D]), cache."""
        c, P = self.cfg, self.p
        x = np.asarray(x)
        if past is not None:
            x = x[:, -1:]                                            # :735-737
            if token_type_ids is not None:
                token_type_ids = np.asarray(token_type_ids)[:, -1:]  # :741-742
        x = x.astype(np.int64)                                       # :758
        B, T = x.shape
        past_len = 0 if past is None else past[0].shape[-2]          # :760-765
        if position_ids is None:
            pos = np.arange(past_len, T + past_len)[None]            # :770-773
        else:
            pos = np.asarray(position_ids).astype(np.int64)
            pos = pos.reshape(-1, pos.shape[-1])                     # :784
        if pos.max() >= c.W:
            raise IndexError("position %d outside wpe table (%d rows)" % (pos.max(), c.W))
        R, Wt = self.R, self._w
        h = P["wte/weight"][x] + P["wpe/embeddings"][pos]            # :137-138,786,793
        if token_type_ids is not None:                               # :787-793
            tt = np.asarray(token_type_ids).astype(np.int64)
            h = h + P["wte/weight"][tt.reshape(-1, tt.shape[-1])]
        h, m_emb = self._dropout(h, c.p_resid, step, 0, 0, training)  # :794
        h = R(h)
        cache = {"x": x, "pos": pos[0] if position_ids is None else pos, "m_emb": m_emb, "layers": [], "hidden": []}
        presents = []
        for i in range(c.L):
            pre = "decoder_blocks/%d/" % i
            lc = {}
            x_in = h
            cache["hidden"].append(h)                                 # all_hidden_states :800-802
            # DecoderBlock.call :574-597 -- NOTE ln_1 output OVERWRITES the residual stream (:583-587)
            if c.use_ln:
                u, lc["ln1"] = layernorm_fwd(x_in, P[pre + "ln_1/gamma"], P[pre + "ln_1/beta"], c.eps)
                u = R(u)
            else:
                u = x_in
            # Attention.call :397-448
            qkv = u.reshape(-1, c.E) @ Wt(pre + "attn/c_attn/weight") + P[pre + "attn/c_attn/bias"]  # :205-209
            qkv = R(qkv).reshape(B, T, 3 * c.E)
            q, k, v = np.split(qkv, 3, axis=2)                        # :417
            sh = lambda t: t.reshape(B, T, c.H, c.D).transpose(0, 2, 1, 3)   # :385-395
            q, k, v = sh(q), sh(k), sh(v)
            if past is not None:                                      # :423-426
                k = np.concatenate([past[i][0], k], axis=-2)
                v = np.concatenate([past[i][1], v], axis=-2)
            presents.append(np.stack([k, v], axis=0))                 # :435
            w = q @ k.transpose(0, 1, 3, 2)                           # :339
            if c.scale:
                w = w * (1.0 / math.sqrt(c.D))                        # :345-348
            nd, ns = w.shape[-2:]
            b = causal_mask(nd, ns, self.dtype)[None, None]
            w = w * b - 1e4 * (1 - b)                                 # :351-354
            if attention_mask is not None:                            # :774-779, 356-358
                w = w + ((1.0 - np.asarray(attention_mask, dtype=self.dtype)) * -10000.0)[:, None, None, :]
            w = w - w.max(-1, keepdims=True)
            pun = np.exp(w)
            lsum = pun.sum(-1, keepdims=True)
            pr = pun / lsum                                           # :360
            prd, m_att = self._dropout(pr, c.p_attn, step, i, 1, training)   # :361
            cache.setdefault("attentions", []).append(prd)            # all_attentions :808-809
            if self.emulate_bf16:
                # the kernel feeds the UNNORMALISED (masked) probabilities to the matrix cores in bf16 and applies
                # keep-scale / row sum to the fp32 output
                pm = pun if m_att is None else pun * (m_att > 0)
                a = (R(pm) @ v) * ((1.0 / (1.0 - c.p_attn)) if m_att is not None else 1.0) / lsum
                a = R(a)
            else:
                a = prd @ v                                           # :367
            a = a.transpose(0, 2, 1, 3).reshape(B, T, c.E)            # :373-383
            ao = a.reshape(-1, c.E) @ Wt(pre + "attn/c_proj/weight") + P[pre + "attn/c_proj/bias"]  # :443
            ao = ao.reshape(B, T, c.E)
            ao, m_ao = self._dropout(ao, c.p_resid, step, i, 2, training)    # :444
            r = R(u + ao)                                             # :587
            if c.use_ln:
                n, lc["ln2"] = layernorm_fwd(r, P[pre + "ln_2/gamma"], P[pre + "ln_2/beta"], c.eps)  # :591
                n = R(n)
            else:
                n = r
            fc = n.reshape(-1, c.E) @ Wt(pre + "mlp/c_fc/weight") + P[pre + "mlp/c_fc/bias"]      # :504
            g = R(gelu(fc))
            fc = R(fc)                                                # the stored pre-activation (backward only)
            mo = g @ Wt(pre + "mlp/c_proj/weight") + P[pre + "mlp/c_proj/bias"]                    # :505
            mo = mo.reshape(B, T, c.E)
            mo, m_mo = self._dropout(mo, c.p_resid, step, i, 3, training)    # :506
            h = R(r + mo)                                             # :594
            if keep_cache:
                lc.update(u=u, q=q, k=k, v=v, pr=pr, m_att=m_att, prd=prd, a=a, m_ao=m_ao, r=r, n=n,
                          fc=fc, g=g, m_mo=m_mo)
                cache["layers"].append(lc)
        hf, lnf = layernorm_fwd(h, P["ln_f/gamma"], P["ln_f/beta"], c.eps)   # :811 (always applied)
        hf = R(hf)
        logits = hf @ Wt("wte/weight").T                              # :139-144,818
        cache["hf"], cache["lnf"] = hf, lnf
        cache["hidden"].append(hf)                                    # ... and the last hidden state :814-816
        return logits, presents, cache

    # ---------------------------------------------------------------- loss
    @staticmethod
    def loss_acc(logits, y):
        """SparseCategoricalCrossentropy(from_logits=True), SUM_OVER_BATCH_SIZE (transformer.py:888,918)
        and batch accuracy (:924-926, argmax lowest index on ties)."""
        V = logits.shape[-1]
        z = logits.reshape(-1, V)
        y = np.asarray(y).reshape(-1).astype(np.int64)
        zmax = z.max(-1, keepdims=True)
        lse = zmax[:, 0] + np.log(np.exp(z - zmax).sum(-1))
        nll = lse - z[np.arange(z.shape[0]), y]
        acc = (z.argmax(-1) == y).mean()
        return nll.mean(), acc

    def loss_and_grads(self, x, y, training=True, step=0):
        """GradientTape over Transformer.call + loss (transformer.py:916-920), written out by hand
        (SURVEY appendix A)."""
        c, P = self.cfg, self.p
        logits, _, cache = self.forward(x, training=training, step=step, keep_cache=True)
        B, T = cache["x"].shape
        N = B * T
        loss, acc = self.loss_acc(logits, y)
        z = logits.reshape(N, c.V)
        yy = np.asarray(y).reshape(-1).astype(np.int64)
        sm = np.exp(z - z.max(-1, keepdims=True))
        sm /= sm.sum(-1, keepdims=True)
        dz = sm
        dz[np.arange(N), yy] -= 1.0
        dz /= N
        R, Wt = self.R, self._w
        dz = R(dz)
        G = {k: np.zeros_like(v) for k, v in P.items()}
        hf = cache["hf"].reshape(N, c.E)
        G["wte/weight"] += dz.T @ hf                                  # tied logits wgrad
        dhf = R(dz @ Wt("wte/weight")).reshape(B, T, c.E)
        dh, G["ln_f/gamma"], G["ln_f/beta"] = layernorm_bwd(dhf, cache["lnf"], P["ln_f/gamma"])
        dh = R(dh)
        sc = (1.0 / math.sqrt(c.D)) if c.scale else 1.0
        for i in reversed(range(c.L)):
            pre = "decoder_blocks/%d/" % i
            lc = cache["layers"][i]
            dx_out = dh
            dmo = dx_out if lc["m_mo"] is None else R(dx_out * lc["m_mo"])
            dmo2 = dmo.reshape(N, c.E)
            G[pre + "mlp/c_proj/weight"] = lc["g"].T @ dmo2
            G[pre + "mlp/c_proj/bias"] = dmo2.sum(0, keepdims=True)
            dg = dmo2 @ Wt(pre + "mlp/c_proj/weight").T
            dfc = dg * gelu_grad(lc["fc"])
            G[pre + "mlp/c_fc/bias"] = dfc.sum(0, keepdims=True)        # from the fp32 accumulators (fused column sums)
            dfc = R(dfc)
            G[pre + "mlp/c_fc/weight"] = lc["n"].reshape(N, c.E).T @ dfc
            dn = R(dfc @ Wt(pre + "mlp/c_fc/weight").T).reshape(B, T, c.E)
            if c.use_ln:
                dln, G[pre + "ln_2/gamma"], G[pre + "ln_2/beta"] = layernorm_bwd(dn, lc["ln2"], P[pre + "ln_2/gamma"])
            else:
                dln = dn
            dr = R(dx_out + dln)
            dao = dr if lc["m_ao"] is None else R(dr * lc["m_ao"])
            dao2 = dao.reshape(N, c.E)
            G[pre + "attn/c_proj/weight"] = lc["a"].reshape(N, c.E).T @ dao2
            G[pre + "attn/c_proj/bias"] = dao2.sum(0, keepdims=True)
            da = R(dao2 @ Wt(pre + "attn/c_proj/weight").T).reshape(B, T, c.H, c.D).transpose(0, 2, 1, 3)
            pr = lc["pr"]
            dprd = da @ lc["v"].transpose(0, 1, 3, 2)
            dpr = dprd if lc["m_att"] is None else dprd * lc["m_att"]
            if self.emulate_bf16:
                # the kernels recompute P in fp32 from the saved row statistics; delta = rowsum(dO * O) on the stored
                # bf16 O; bf16 P (dropped, unscaled) feeds dV and bf16 dS feeds dQ / dK, keep-scale on the fp32 outputs
                ks = (1.0 / (1.0 - c.p_attn)) if lc["m_att"] is not None else 1.0
                keep = 1.0 if lc["m_att"] is None else (lc["m_att"] > 0)
                o_bf = lc["a"].reshape(B, T, c.H, c.D).transpose(0, 2, 1, 3)
                delta = (da * o_bf).sum(-1, keepdims=True)
                dv = ks * (R(pr * keep).transpose(0, 1, 3, 2) @ da)
                dw = pr * (dprd * keep - delta / ks) * causal_mask(T, T, self.dtype)[None, None]
                dwb = R(dw)
                dq = sc * ks * (dwb @ lc["k"])
                dk = sc * ks * (dwb.transpose(0, 1, 3, 2) @ lc["q"])
            else:
                dv = lc["prd"].transpose(0, 1, 3, 2) @ da
                dw = pr * (dpr - (dpr * pr).sum(-1, keepdims=True))
                dw = dw * causal_mask(T, T, self.dtype)[None, None]   # d(w*b)
                dq = sc * (dw @ lc["k"])
                dk = sc * (dw.transpose(0, 1, 3, 2) @ lc["q"])
            mh = lambda t: t.transpose(0, 2, 1, 3).reshape(B, T, c.E)
            dqkv = np.concatenate([mh(dq), mh(dk), mh(dv)], axis=2).reshape(N, 3 * c.E)
            G[pre + "attn/c_attn/bias"] = dqkv.sum(0, keepdims=True)    # from the fp32 accumulators
            dqkv = R(dqkv)
            G[pre + "attn/c_attn/weight"] = lc["u"].reshape(N, c.E).T @ dqkv
            du = R(dr + (dqkv @ Wt(pre + "attn/c_attn/weight").T).reshape(B, T, c.E))
            if c.use_ln:
                dh, G[pre + "ln_1/gamma"], G[pre + "ln_1/beta"] = layernorm_bwd(du, lc["ln1"], P[pre + "ln_1/gamma"])
                dh = R(dh)
            else:
                dh = du
        dh0 = dh if cache["m_emb"] is None else dh * cache["m_emb"]       # masked in fp32 inside the scatter-add kernel
        np.add.at(G["wte/weight"], cache["x"].reshape(-1), dh0.reshape(N, c.E))   # gather grad
        G["wpe/embeddings"][cache["pos"]] += dh0.sum(0)
        return loss, acc, G, logits

    # ---------------------------------------------------------------- optimizer
    def adam_step(self, G, lr, beta1=0.9, beta2=0.999, eps=1e-7):
        """Keras OptimizerV2 Adam dense update (transformer.py:887,921): eps OUTSIDE the bias
        correction: theta -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps).  wpe's sparse path is the
        non-lazy Keras sparse Adam == dense update with zero rows."""
        self.iterations += 1
        t = self.iterations
        alpha = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
        for k in self.p:
            g = G[k]
            self.m[k] = beta1 * self.m[k] + (1 - beta1) * g
            self.v[k] = beta2 * self.v[k] + (1 - beta2) * g * g
            self.p[k] = self.p[k] - alpha * self.m[k] / (np.sqrt(self.v[k]) + eps)

    def train_step(self, x, y, lr, training=True):
        """One iteration of the loop body at transformer.py:914-930."""
        loss, acc, G, _ = self.loss_and_grads(x, y, training=training, step=self.iterations)
        self.adam_step(G, lr)
        return loss, acc

    # ---------------------------------------------------------------- decode
    def generate_literal(self, prompt, n, temperature=0.0, rng=None):
        """Bit-for-bit restatement of cli.py:659-676: `past` is never fed back, so after the first
        step the model sees ONE token at position 0.  temperature 0 => argmax (lowest index)."""
        x = np.asarray(prompt, dtype=np.int64)[None]
        out = []
        for _ in range(n):
            logits, _, _ = self.forward(x)
            out.append(self._sample(logits[0, -1], temperature, rng))
            x = np.array([[out[-1]]])
        return out

    def generate_kv(self, prompt, n, temperature=0.0, rng=None):
        """model(x, past=presents) semantics (transformer.py:735-765,423-426): prompt once, then one
        new token per step at position P+i attending to the whole cache."""
        x = np.asarray(prompt, dtype=np.int64)[None]
        if x.shape[1] + n - 1 > self.cfg.W:
            raise IndexError("prompt_len + length - 1 exceeds window_size")
        logits, past, _ = self.forward(x)
        out = []
        for i in range(n):
            out.append(self._sample(logits[0, -1], temperature, rng))
            if i + 1 < n:
                logits, past, _ = self.forward(np.array([[out[-1]]]), past=past)
        return out

    @staticmethod
    def _sample(z, temperature, rng):
        if temperature <= 0.0:
            return int(np.argmax(z))
        z = z / temperature                                           # cli.py:671
        p = np.exp(z - z.max()); p /= p.sum()
        return int(rng.choice(len(p), p=p))                           # tf.random.categorical, cli.py:673


def synthetic_batch(rng, V, B, T):
    """x, y = seq[:, :-1], seq[:, 1:] -- the shift-by-one of models/__init__.py:304."""
    seq = rng.integers(0, V, size=(B, T + 1), dtype=np.int32)
    return seq[:, :-1].copy(), seq[:, 1:].copy()
