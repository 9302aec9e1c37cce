"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

Second, INDEPENDENT CPU restatement of the reference hot path (galacticglum/composer
`composer/models/transformer.py`) written with torch ops + autograd.  It shares no code with
`oracle/transformer_oracle.py` (numpy, hand-written backward); `tests/test_oracle.py` holds the two
against each other in float64 (agreement ~1e-12), and `bench.py`'s `cpu_baseline` leg times this one in
float32 on all host cores (torch's CPU BLAS + autograd is the closest thing on the box to the reference's
TF-CPU eager path: same op granularity, scores materialised as `[B,H,T,T]` tensors like
transformer.py:339-367).  Only `tests/` and `bench.py`'s cpu_baseline may import this module.

PARITY UNPINNED (same reason as transformer_oracle.py: TensorFlow is not available offline and the
reference's tests hold no vector for this path).
"""
import math

import torch


def forward(P, x, cfg, past_len=0):
    """Transformer.call (transformer.py:696-833) with past=None, dropout off.  P: name -> tensor, x: long [B,T]."""
    B, T = x.shape
    E, H, D = cfg.E, cfg.H, cfg.D
    pos = torch.arange(past_len, past_len + T)
    h = P["wte/weight"][x] + P["wpe/embeddings"][pos]                               # :137-138,786,793
    for i in range(cfg.L):
        p = "decoder_blocks/%d/" % i
        if cfg.use_ln:                                                               # :583-584 (overwrites the stream)
            h = torch.nn.functional.layer_norm(h, (E,), P[p + "ln_1/gamma"], P[p + "ln_1/beta"], cfg.eps)
        qkv = h.reshape(-1, E) @ P[p + "attn/c_attn/weight"] + P[p + "attn/c_attn/bias"]   # :205-209
        q, k, v = qkv.reshape(B, T, 3 * E).split(E, dim=2)                           # :417
        q, k, v = [t.reshape(B, T, H, D).permute(0, 2, 1, 3) for t in (q, k, v)]     # :385-395
        w = q @ k.transpose(-1, -2)                                                  # :339
        if cfg.scale:
            w = w * (1.0 / math.sqrt(D))                                             # :345-348
        b = torch.tril(torch.ones(T, T, dtype=w.dtype))                              # :290-301
        w = w * b - 1e4 * (1 - b)                                                    # :351-354
        w = torch.softmax(w, -1)                                                     # :360
        a = (w @ v).permute(0, 2, 1, 3).reshape(B, T, E)                             # :367,373-383
        a = (a.reshape(-1, E) @ P[p + "attn/c_proj/weight"] + P[p + "attn/c_proj/bias"]).reshape(B, T, E)
        h = h + a                                                                    # :587
        m = h
        if cfg.use_ln:
            m = torch.nn.functional.layer_norm(h, (E,), P[p + "ln_2/gamma"], P[p + "ln_2/beta"], cfg.eps)   # :591
        f = m.reshape(-1, E) @ P[p + "mlp/c_fc/weight"] + P[p + "mlp/c_fc/bias"]     # :504
        f = 0.5 * f * (1 + torch.tanh(math.sqrt(2 / math.pi) * (f + 0.044715 * f ** 3)))   # :35-40
        f = (f @ P[p + "mlp/c_proj/weight"] + P[p + "mlp/c_proj/bias"]).reshape(B, T, E)  # :505
        h = h + f                                                                    # :594
    h = torch.nn.functional.layer_norm(h, (E,), P["ln_f/gamma"], P["ln_f/beta"], cfg.eps)   # :811
    return h @ P["wte/weight"].T                                                     # :139-144,818


class TorchTrainer:
    """The loop body of transformer.py:914-930 (forward, sparse-CE mean, tape gradient, Keras Adam) on torch CPU."""

    def __init__(self, cfg, params, dtype=torch.float32):
        self.cfg = cfg
        self.P = {k: torch.tensor(v, dtype=dtype, requires_grad=True) for k, v in params.items()}
        self.m = {k: torch.zeros_like(v) for k, v in self.P.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.P.items()}
        self.iterations = 0

    def train_step(self, x, y, lr, beta1=0.9, beta2=0.999, eps=1e-7):
        x = torch.as_tensor(x, dtype=torch.long)
        y = torch.as_tensor(y, dtype=torch.long)
        logits = forward(self.P, x, self.cfg)
        loss = torch.nn.functional.cross_entropy(logits.reshape(-1, self.cfg.V), y.reshape(-1))   # :888,918
        grads = torch.autograd.grad(loss, list(self.P.values()))                                  # :916-920
        self.iterations += 1
        t = self.iterations
        alpha = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)                             # Keras Adam, :887,921
        with torch.no_grad():
            for (k, p), g in zip(self.P.items(), grads):
                self.m[k].mul_(beta1).add_(g, alpha=1 - beta1)
                self.v[k].mul_(beta2).addcmul_(g, g, value=1 - beta2)
                p.sub_(alpha * self.m[k] / (self.v[k].sqrt() + eps))
            acc = (logits.argmax(-1) == y).float().mean().item()
        return float(loss.item()), acc
