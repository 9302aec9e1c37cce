#!/usr/bin/env python3
"""Which part of tools/decode_bench.py crashes rocprofv3 on this image:  rocprofv3 --kernel-trace -- python3 tools/decode_rocprof_probe.py <case>
Measured (round 3): graph (kv, 2 x 256 tokens) ok | eager ok | switch (graph then eager: state rebuilt) ok | two_eager (kv then literal, eager) ok |
lit_short (literal, graph, 8 tokens) ok | lit (literal, graph, 2 x 256 tokens) SIGSEGV | two (kv then literal, graph) SIGSEGV -- a memcpy running
off the end of a mapping six frames inside the profiler's tool library, under hipGraphLaunch called from cmp_decode_steps.  Without the profiler
every case runs (and the fuzzers / soak probe replay literal-mode graphs thousands of times); tools/profile_round3.sh therefore takes the decode
chain's per-kernel numbers from the eager trace."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from composer_amd.transformer import Transformer
what = sys.argv[1]
V, E, H, L, W = 390, 512, 8, 6, 2048
m = Transformer(V, E, W, L, H, attention_dropout_rate=0.0, residual_dropout_rate=0.0, dtype="fp32", seed=0, max_batch=1, max_seq=64)
prompt = np.random.default_rng(0).integers(0, V, 10)
def gen(graph, n=256, mode="kv"):
    os.environ["COMPOSER_NO_GRAPH"] = "0" if graph else "1"
    return m.generate(prompt, n, temperature=1.0, mode=mode, seed=1)
if what == "graph":
    gen(True); gen(True)
elif what == "eager":
    gen(False); gen(False)
elif what == "switch":
    gen(True); gen(False)
elif what == "two":
    gen(True, mode="kv"); gen(True, mode="literal")
elif what == "two_eager":
    gen(False, mode="kv"); gen(False, mode="literal")
elif what == "lit":
    gen(True, mode="literal"); gen(True, mode="literal")
elif what == "lit_short":
    gen(True, n=8, mode="literal")
print(what, "ok")
