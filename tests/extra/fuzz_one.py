#!/usr/bin/env python3
"""One configuration of fuzz_shapes.py's domain, per-parameter gradient errors printed (test infrastructure, GPU box only):
    python tests/extra/fuzz_one.py "{'V': 2, 'E': 320, ...}" """
import ast, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import transformer_oracle as O
from composer_amd import _lib
from composer_amd.transformer import Transformer
c = ast.literal_eval(sys.argv[1])
V, E, H, L, W, T, B = c["V"], c["E"], c["H"], c["L"], c["W"], c["T"], c["B"]
rng = np.random.default_rng(c["seed"])
std = float(os.environ.get("FUZZ_STD", "0.1"))
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
m.set_weights(params)
loss, acc, G, _ = orc.loss_and_grads(x, y, training=c["p"] > 0, step=0)
l2, a2 = m.loss_and_grads(x, y)
print("loss", l2, loss)
rows = []
for n in m.parameter_names:
    gr = m.get_parameter(n, _lib.KIND_GRAD).astype(np.float64)
    rows.append((np.abs(gr - G[n]).max() / (np.abs(G[n]).max() + 1e-5), n, np.abs(G[n]).max()))
for r in sorted(rows, reverse=True)[:6]:
    print("%.4f  %-28s  max|g| %.3e" % r)
m.close()
