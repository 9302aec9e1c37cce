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


def test_bench_under_the_launcher_reports_the_gradient_exchange():
    """`bench.py --gpus 1` under torch.distributed.run takes the product's RCCL path (one-rank communicator,
    per-block buckets on the priority stream, Adam per bucket) and its JSON line carries `comm`; the default N=1 run embeds the same
    through a child process as `dp1` (bench.py: dp1_child) -- here the launched form itself, small and quick."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
                        "--batch", "32", "--no-extras", "--no-cpu-baseline", "--no-decode"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
    doc = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    comm = doc["comm"]
    assert comm["ranks"] == 1 and comm["steps"] == 4 and comm["buckets"] == 6 + 2 + 1          # L blocks + ln_f + embeddings, + the metrics message
    assert comm["bytes"] > 4 * 19e6 * 0.9                                                          # every fp32 gradient once
    assert 0.0 <= comm["exposed_ms"] / comm["steps"] < 0.5 * doc["ms_per_step"]
    # round 6: the line of a launched run is self-diagnosing -- every rank's own step time and exposed communication, and which RCCL /
    # HIP runtime the process is bound to (one copy of each: _lib.check_single_runtime); stdout holds the JSON line and nothing else
    assert [r["rank"] for r in doc["ranks"]] == [0] and doc["ranks"][0]["ms_per_step"] > 0 and doc["ranks"][0]["exposed_ms"] is not None
    rt = doc["runtime"]
    assert rt["rccl_version"] >= 20000 and os.path.exists(rt["rccl_path"]) and os.path.exists(rt["hip_runtime_path"])
    assert [l for l in p.stdout.splitlines() if l.strip()] == [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_bench_allreduce_only_times_the_products_bucket_pattern():
    """`bench.py --allreduce-only`: the gradient exchange of one step alone (cmp_dp_allreduce_pattern) -- 3-float metrics message +
    ln_f, L blocks and the embeddings as fp32 buckets, every gradient byte once -- under the launcher with one rank (RCCL copies in
    place; N > 1 needs a multi-GPU node).  The first thing to run when an 8-GPU job scales badly."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--allreduce-only", "--steps", "5"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
    doc = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert doc["n_gpus"] == 1 and doc["messages"] == 6 + 2 + 1 and doc["reps"] == 5
    assert doc["bytes"] == 4 * 19639296 + 12                                  # C2 has 19 639 296 parameters
    assert doc["ms_per_pattern"] > 0 and doc["value"] > 1.0 and doc["runtime"]["rccl_version"] >= 20000
