"""-m gpu: the ctypes binding printed in INTEGRATION.md section 2 (what a maintainer of the reference would paste) is executed
as written -- extracted from the markdown -- against libcomposer_hip.so, and its functions are checked against the host class."""
import os
import re
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_integration_md_stub_runs_as_written():
    from composer_amd import _lib
    from composer_amd.transformer import Transformer
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = md[md.index("## 2. Bind the C ABI directly"):]
    code = re.search(r"```python\n(.*?)```", sec, re.S).group(1)
    code = code.replace('C.CDLL("libcomposer_hip.so")', 'C.CDLL(%r)' % _lib.LIB_PATH)
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)          # creates a ctx and the default-config model (E=256, L=8, H=16, bf16)
    ref = Transformer(390, 256, 1024, 8, 16, dtype="bf16", seed=0, max_batch=1, max_seq=1024)
    for n in ref.parameter_names:                               # same weights on both sides
        ns["set_param"](n, ref.get_parameter(n))
    rng = np.random.default_rng(0)
    x = rng.integers(0, 390, (1, 64)); y = rng.integers(0, 390, (1, 64))
    want, presents = ref(x)
    got = ns["logits"](x)
    assert got.shape == (1, 64, 390) and np.abs(got - want).max() <= 2e-2
    # past: presents of the first 63 positions + the last token reproduce the last row
    _, p63 = ref(x[:, :63])
    lp = ns["logits_with_past"](x[:, 63:64], [np.array(p) for p in p63])
    assert np.abs(lp[:, 0] - want[:, 63]).max() <= 3e-2
    l1, a1 = ns["train_step"](x, y, 1e-3)
    l2, a2 = ref.train_step(x, y, 1e-3)
    assert abs(l1 - l2) <= 2e-2 * abs(l2) and np.isfinite(l1)
    out = list(ns["train_pipelined"]([(x, y)] * 4, 1e-3))
    assert len(out) == 3 and all(np.isfinite(l) for l, _ in out) and out[-1][0] < l1
    ids = ns["generate"](x[0, :10], 16, temperature=0.0)
    assert len(ids) == 16 and ((ids >= 0) & (ids < 390)).all()
    ref.close()
