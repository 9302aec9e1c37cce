#!/usr/bin/env python3
"""bench.py -- throughput of the Transformer teacher-forced TRAIN STEP (forward + sparse-CE + backward +
[RCCL all-reduce] + Adam) on synthetic int32 MIDI-event sequences, BASELINE.json's metric:
"MIDI-event tokens/sec (train, seq=1024)".

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c4]

With --gpus N > 1 and no launcher environment, this process starts `python -m torch.distributed.run` with N ranks of
itself (before anything touches the GPU), forwards rank 0's JSON line and exits with the launcher's status; under a
launcher (RANK/WORLD_SIZE set) it is one of the ranks.

Workloads: c2 (default; BASELINE configs 2/3) 6L/8H/d512, window 1024, B=128 sequences per GPU;
c4 (BASELINE config 4) 12L/12H/d768, window 2048, B=32 per GPU.  Weak scaling (--batch is per GPU).  bf16 activations
with fp32 master weights/accumulation, dropout 0.1 (default_config.yml:39-40), Adam lr 1e-3.
Inputs are generated up front and live in HBM before the timed region.

One JSON line on rank 0 with `roofline` (live HIP-event timing of the dominant kernel class inside the timed
region), `cpu_baseline` (the CPU restatement of the reference path on the host cores, bounded sample; N=1 only),
`decode` (BASELINE config 5, with its own memory roofline) and, at N=1, more driver-timed objects: `classes` (the live
roofline of EVERY kernel class of the step, three extra steps each; every entry carries `algorithmic_bytes` and, when
profiles/hbm_traffic.json covers the run, the PMC-measured in-step `traffic`), `b32` (SURVEY's C2 batch, 32 sequences, with its
own `classes`) and `c4` (BASELINE config 4), each `{ms_per_step, value, model_mfma_frac}` from 5 warm-up + 20 timed steps, and
`forward` (the north-star quantity: the inference forward pass of C2 and C4 as a fraction of the bf16 MFMA peak).
Under a launcher (any N, also 1) the line carries `comm`: the gradient exchange's exposed (non-overlapped) time per step, `ranks`
(every rank's own ms/step and exposed ms, gathered over gloo -- a straggler shows up by rank) and `runtime` (RCCL version and the
paths of the RCCL / HIP runtime this process is bound to); RCCL's own diagnostics (NCCL_DEBUG, default WARN) go to stderr: under
a launcher fd 1 points at stderr for the whole run and the JSON line is written to the original stdout.

  python bench.py --gpus N --allreduce-only [--steps K]
times the gradient exchange of one step ALONE -- the product's own message pattern (3-float metrics message + L+2 fp32 gradient
buckets, 78.6 MB at C2) back to back on the communication stream, no compute -- and prints algorithm / bus bandwidth.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# RCCL / cross-process GPU memory sharing on this pool needs dmabuf IPC (the image exports it; keep it if a launcher dropped it)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

V = 390
CONFIGS = {
    # name: (E, H, L, W=T, default B per GPU; not fixed by BASELINE -- sweep in DESIGN.md section 7)
    "c2": dict(E=512, H=8, L=6, T=1024, B=128, label="6L/8H/d512"),
    "c4": dict(E=768, H=12, L=12, T=2048, B=32, label="12L/12H/d768"),
}
LR = 1e-3
PEAK_BF16_TFLOPS = 2500.0       # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
KERNEL_CLASSES = {
    0: ("gemm_bf16_256_kernel<A[M,K], W^T[N,K]> (forward Conv1D: c_attn/c_proj/c_fc/mlp c_proj, tied logits)", "mfma"),
    1: ("gemm_bf16_256_kernel<A[M,N], W[K,N]> (dgrad)", "mfma"),
    2: ("gemm_bf16_p4_kernel<A stored [K,M]> (wgrad, split-K f32 atomics)", "mfma"),
    3: ("attn_fwd_kernel<bf16,64>", "mfma"),
    4: ("attn_dq_kernel<bf16,64>", "mfma"),
    5: ("attn_dkv_kernel<bf16,64>", "mfma"),
    6: ("layernorm_fwd_kernel<bf16>", "hbm"),
    7: ("adam_kernel", "hbm"),
    8: ("layernorm_bwd_kernel<bf16> (+ residual add, masked copy, bias-gradient column sums)", "hbm"),
}


def pmc_traffic(kernel_class, tokens, cfg_name):
    """HBM-side bytes per launch of a kernel class INSIDE THE TRAIN STEP, from the committed PMC passes over this very script
    (profiles/hbm_traffic.json, made by tools/profile_round4.sh + tools/make_traffic_json.py: `rocprofv3 --pmc FETCH_SIZE` and
    `--pmc WRITE_SIZE` in separate runs of `python3 bench.py --no-extras --no-cpu-baseline --no-decode`, the gfx950 FETCH_SIZE
    correction applied).  Counters cannot be read from inside this process; None when the file does not cover the run."""
    return pmc_traffic_note(kernel_class, tokens, cfg_name)[0]


def pmc_traffic_note(kernel_class, tokens, cfg_name):
    """(bytes per launch or None, why).  A run of the file counts only while the library it was measured on (its `build_key`, written by
    tools/make_traffic_json.py) is the library loaded now: a kernel change makes the committed counters stale, and stale counters
    are reported as None, never as current."""
    try:
        doc = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    except Exception:
        return None, "profiles/hbm_traffic.json is missing or unreadable"
    run = doc.get("runs", {}).get("%s_tokens%d" % (cfg_name, tokens))
    if not run:
        return None, "profiles/hbm_traffic.json has no PMC run for %s at %d tokens" % (cfg_name, tokens)
    from composer_amd import _lib
    key = _lib.load().cmp_build_key().decode()
    if run.get("build_key") != key:
        return None, ("profiles/hbm_traffic.json was measured on library build %s..., the library loaded is %s...: stale counters are not "
                      "printed (re-run tools/profile_round5.sh)" % (str(run.get("build_key"))[:12], key[:12]))
    ent = run.get("classes", {}).get(str(kernel_class))
    if not ent:
        return None, "class %d not in the PMC run" % kernel_class
    return ent.get("traffic_bytes_per_launch"), ("HBM-side bytes per launch of this class inside the train step (mean over its launches): "
                                                  "rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, separate passes over bench.py itself on this "
                                                  "library build, profiles/hbm_traffic.json")


def class_table(lib, step, tokens, cfg_name, reps=3):
    """Live roofline of every kernel class of the step: HIP events around each launch of the class on its own stream
    (cmp_prof_*), `reps` steps per class, outside the timed region."""
    rows = []
    for cls in sorted(KERNEL_CLASSES):
        lib.cmp_prof_begin(cls)
        for i in range(reps):
            step(i)
        cms, cn, cw, cb = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        lib.cmp_prof_end2(C.byref(cms), C.byref(cn), C.byref(cw), C.byref(cb))
        if cn.value <= 0 or cms.value <= 0:
            continue
        nm, bound = KERNEL_CLASSES[cls]
        rate = cw.value / (cms.value * 1e-3)
        peak = PEAK_BF16_TFLOPS * 1e12 if bound == "mfma" else PEAK_HBM_GBS * 1e9
        row = {"class": cls, "kernel": nm.split(" (")[0], "bound": bound, "launches_per_step": cn.value // reps,
               "avg_us": 1e3 * cms.value / cn.value, "ms_per_step": cms.value / reps,
               "achieved": rate / (1e12 if bound == "mfma" else 1e9), "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
               "frac": rate / peak, "algorithmic_bytes": cb.value / cn.value, "traffic": pmc_traffic(cls, tokens, cfg_name)}
        # (None when profiles/hbm_traffic.json was measured on another build of the library: pmc_traffic_note)
        if row["traffic"]:
            row["traffic_over_algorithmic"] = row["traffic"] / row["algorithmic_bytes"]
            row["hbm_gbs"] = row["traffic"] / (row["avg_us"] * 1e-6) / 1e9
        rows.append(row)
    return rows


def forward_bench(device):
    """north_star: "fraction of the bf16 MFMA peak on the attention+FFN forward at seq=1024".  The inference forward pass
    (cmp_eval_step: embed -> L blocks -> ln_f -> tied logits -> loss; dropout off, the ids upload and the call's sync included,
    < 1 %) of C2 at B=128 and C4 at B=32, median of 20 calls.  FLOPs per token = L*(24E^2 + 2ET) + 2EV (causal attention on the
    unmasked half); `frac_attn_ffn` leaves the logits GEMM's FLOPs out of the numerator (time unchanged)."""
    from composer_amd.transformer import Transformer
    out = {}
    for name in ("c2", "c4"):
        cf = CONFIGS[name]
        E, H, L, T, B = cf["E"], cf["H"], cf["L"], cf["T"], cf["B"]
        m = Transformer(V, E, T, L, H, attention_dropout_rate=0.0, residual_dropout_rate=0.0, dtype="bf16", seed=0, max_batch=B,
                        max_seq=T, device=device)
        m.initialize_parameters(0)
        rng = np.random.default_rng(0)
        ds = [(rng.integers(0, V, (B, T), dtype=np.int32), rng.integers(0, V, (B, T), dtype=np.int32))]
        for _ in range(3):
            m.evaluate(ds)
        ts = []
        for _ in range(20):
            t0 = time.perf_counter()
            m.evaluate(ds)
            ts.append(time.perf_counter() - t0)
        # the decoder-block stack alone (HIP events around the L blocks inside the same call, timing class 9): embedding, ln_f,
        # tied logits, loss, metrics and the host side of the call left out -- the span north_star's "attention+FFN forward" names
        from composer_amd import _lib
        lib = _lib.load()
        bts = []
        for _ in range(10):                                    # per call, median (like `ms` above)
            lib.cmp_prof_begin(9)
            m.evaluate(ds)
            bms, bn, bw, bb = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
            lib.cmp_prof_end2(C.byref(bms), C.byref(bn), C.byref(bw), C.byref(bb))
            if bn.value == 1:
                bts.append(bms.value)
        fused, _n = C.c_int(-1), C.c_int64(0)
        lib.cmp_model_path_info(m._h, C.byref(fused), C.byref(_n))
        m.close()
        dt = float(np.median(ts))
        blocks = L * (24 * E * E + 2 * E * T)
        tf = B * T * (blocks + 2 * E * V) / dt / 1e12
        out[name] = {"workload": "%s inference forward, seq=%d, B=%d, bf16" % (cf["label"], T, B), "ms": 1e3 * dt,
                     "tokens_per_s": B * T / dt, "tflops": tf, "frac": tf / PEAK_BF16_TFLOPS,
                     "frac_attn_ffn": B * T * blocks / dt / 1e12 / PEAK_BF16_TFLOPS,
                     "layernorm_fused_block_path": bool(fused.value == 1)}
        if bts:
            bdt = float(np.median(bts)) * 1e-3
            out[name]["blocks_only_ms"] = 1e3 * bdt
            out[name]["frac_attn_ffn_blocks_only"] = B * T * blocks / bdt / 1e12 / PEAK_BF16_TFLOPS
    return out


class stdout_to_stderr:
    """gloo and RCCL print connection / version banners on fd 1 while they initialise; the contract is ONE JSON line on
    stdout, so fd 1 points at stderr for the duration."""
    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def flops_per_token_train(E, L, T):
    fwd = L * (24 * E * E + 2 * E * T) + 2 * E * V          # causal attention counted on the unmasked half
    return 3 * fwd


def cpu_baseline(cf, seconds_budget=24.0):
    """The reference path restated for the CPU, timed on the host cores: train steps (fwd + bwd + Keras Adam, float32,
    dropout off) of (a) the torch restatement (oracle/torch_restatement.py: torch CPU ops + autograd, scores materialised
    like the reference's TF eager path) and (b) the numpy oracle, at the batch that fits the time budget; the faster of
    the two is reported."""
    from oracle import transformer_oracle as O          # cpu_baseline leg only
    E, H, L, T = cf["E"], cf["H"], cf["L"], cf["T"]
    cfg = O.Config(V, E, T, L, H)
    params = O.init_params(V, E, T, L, seed=0, dtype=np.float32)
    rng = np.random.default_rng(1234)
    results = []

    def run(name, make, Bc, budget, cores):
        eng = make()
        x, y = O.synthetic_batch(rng, V, Bc, T)
        t0 = time.time()
        eng(x, y)                                       # warm-up (thread pools, page faults)
        warm = time.time() - t0
        n, t0 = 0, time.time()
        while True:
            eng(x, y)
            n += 1
            if time.time() - t0 > max(1.5, budget - warm) or n >= 12:
                break
        dt = time.time() - t0
        results.append({"value": Bc * T * n / dt, "unit": "tokens/s", "cores": int(cores), "kind": "port",
                        "sample": "%d train steps (fwd+bwd+Adam, float32, dropout off) of the %s at B=%d, T=%d, %s"
                                  % (n, name, Bc, T, cf["label"])})

    try:
        import torch
        from oracle.torch_restatement import TorchTrainer
        # torch's CPU ops stop scaling (and then collapse) well below the box's 128-256 hardware threads on this op mix:
        # measured on the GPU box (tests/extra/cpu_baseline_sweep.py) 16 threads x B=1 is the fastest, 128 threads 10x slower.
        ncpu = os.cpu_count() or 1
        for th in sorted({min(16, ncpu), min(32, ncpu)}):
            torch.set_num_threads(th)

            def mk():
                tr = TorchTrainer(cfg, params)
                return lambda x, y: tr.train_step(x, y, LR)
            run("torch-CPU restatement of transformer.py (%d threads)" % th, mk, 1, seconds_budget * 0.4, th)
    except Exception as e:                              # torch CPU ops unavailable: numpy leg only
        print("cpu_baseline: torch leg failed: %r" % (e,), file=sys.stderr)
    try:
        from threadpoolctl import threadpool_info
        ncores = max([i.get("num_threads", 1) for i in threadpool_info()] + [1])
    except Exception:
        ncores = os.cpu_count() or 1

    def mk_np():
        orc = O.OracleTransformer(cfg, params, dtype=np.float32)
        return lambda x, y: orc.train_step(x, y, LR, training=False)
    run("numpy oracle", mk_np, 1, seconds_budget * 0.2, ncores)
    best = max(results, key=lambda r: r["value"])
    best["others"] = [{"value": r["value"], "sample": r["sample"]} for r in results if r is not best]
    return best


def decode_bench(device):
    """BASELINE config 5: generate 1024 tokens at temperature 1.0 from a 10-id prompt, KV cache + hipGraph per-token
    step, the 6L/8H/d512 model with window 2048 (prompt + length must fit the wpe table).  Roofline: the bytes one
    token step must read (fp32 decode weights: every parameter except the unused wpe rows, plus the K/V cache rows up
    to the current position) / time, against the HBM peak (the set fits the 256 MiB Infinity Cache)."""
    from composer_amd.transformer import Transformer
    E, H, L, W = 512, 8, 6, 2048
    P0, N = 10, 1024
    m = Transformer(V, E, W, L, H, attention_dropout_rate=0.0, residual_dropout_rate=0.0, dtype="fp32", seed=0,
                    max_batch=1, max_seq=64, device=device)
    prompt = np.random.default_rng(0).integers(0, V, P0)
    m.generate(prompt, 32, temperature=1.0, mode="kv", seed=1)               # warm-up
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        m.generate(prompt, N, temperature=1.0, mode="kv", seed=1)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    nparam = sum(int(np.prod(m.parameter_shape(n))) for n in m.parameter_names)
    m.close()
    weight_bytes = 4 * (nparam - W * E + E)
    mean_pos = P0 + (N - 1) / 2.0
    kv_bytes = 4 * (2 * L * mean_pos * E + 2 * L * E)
    bpt = weight_bytes + kv_bytes
    gbs = bpt / (best / N) / 1e9
    # SURVEY 8d's definition of the same quantity: bf16 weights excluding the unused wpe rows + bf16 K/V -- half the bytes this
    # fp32 decode chain (the parity mode: bit-exact greedy ids) actually reads; both fractions are reported
    bpt16 = bpt / 2
    gbs16 = bpt16 / (best / N) / 1e9
    return {"metric": "decode tokens/sec (generate len=1024, temp=1.0, batch 1, KV cache + hipGraph)",
            "value": N / best, "unit": "tokens/s", "us_per_token": 1e6 * best / N, "dtype": "f32",
            "launches_per_token": 5 * L + 2,
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                         "bytes_per_token": bpt, "traffic": None,
                         "bytes_per_token_bf16_definition": bpt16, "achieved_bf16_definition": gbs16,
                         "frac_bf16_definition": gbs16 / PEAK_HBM_GBS,
                         "note": "fp32 weights %.1f MB + mean K/V cache read %.2f MB per token; the chain is %d dependent launches "
                                 "(LN1+c_attn, attention, combine+c_proj, LN2+c_fc+GELU, mlp c_proj per block; LN_f+logits; sampler), "
                                 "launch-boundary bound, not byte bound" % (weight_bytes / 1e6, kv_bytes / 1e6, 5 * L + 2)}}


def side_config(name, Bq, device, dropout, steps=20, warmup=5, classes=False):
    """A short timed run of another configuration on the same GPU (N=1): 5 warm-up + 20 steps, inputs resident in HBM.  (Round 6: 3 + 10
    steps read 7.82 ms for C2 at B=32 once where the same process on the same box reads 7.25 on every repeat, tools/b32_probe.py.)"""
    import torch
    from composer_amd.transformer import Transformer
    cf = CONFIGS[name]
    E, H, L, T = cf["E"], cf["H"], cf["L"], cf["T"]
    m = Transformer(V, E, T, L, H, attention_dropout_rate=dropout, residual_dropout_rate=dropout, dtype="bf16", seed=1000,
                    max_batch=Bq, max_seq=T, device=device)
    m.initialize_parameters(0)
    rng = np.random.default_rng(4321)
    seq = rng.integers(0, V, size=(2, Bq, T + 1), dtype=np.int32)
    dev = torch.device("cuda", device)
    xs = [torch.from_numpy(np.ascontiguousarray(seq[i, :, :-1])).to(dev) for i in range(2)]
    ys = [torch.from_numpy(np.ascontiguousarray(seq[i, :, 1:])).to(dev) for i in range(2)]
    for i in range(warmup):
        m.train_step_device(xs[i % 2].data_ptr(), ys[i % 2].data_ptr(), Bq, T, LR)
    m.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        m.train_step_device(xs[i % 2].data_ptr(), ys[i % 2].data_ptr(), Bq, T, LR)
    m.synchronize()
    dt = time.perf_counter() - t0
    loss, _ = m.last_metrics()
    table = None
    if classes:
        from composer_amd import _lib
        table = class_table(_lib.load(), lambda i: m.train_step_device(xs[i % 2].data_ptr(), ys[i % 2].data_ptr(), Bq, T, LR),
                            Bq * T, name)
    m.close()
    value = Bq * T * steps / dt
    out = {"workload": "%s, seq=%d, B=%d, dropout %.2f" % (cf["label"], T, Bq, dropout), "steps": steps, "warmup": warmup,
           "ms_per_step": 1e3 * dt / steps, "value": value, "unit": "tokens/s",
           "model_mfma_frac": value * flops_per_token_train(E, L, T) / 1e12 / PEAK_BF16_TFLOPS, "final_loss": loss}
    if table is not None:
        out["classes"] = table
    return out


def default_config_bench(device, steps=60, warmup=10):
    """The reference's DEFAULT workload (composer/default_config.yml:32-48, what `composer train` runs with no -c): E=256, 16 heads
    of 16, 8 blocks, window 1024, batch_size 1, dropout 0.1 -- 1 024 tokens per step, every kernel latency-bound.  Device-pointer steps
    back to back; `launches` = kernel launches one step enqueues (cmp_train_step_launches: counted on a dropped stream capture)."""
    import torch
    from composer_amd.transformer import Transformer
    E, H, L, T, B = 256, 16, 8, 1024, 1
    m = Transformer(V, E, T, L, H, attention_dropout_rate=0.1, residual_dropout_rate=0.1, dtype="bf16", seed=0, max_batch=B, max_seq=T,
                    device=device)
    m.initialize_parameters(0)
    rng = np.random.default_rng(7)
    seq = rng.integers(0, V, size=(B, T + 1), dtype=np.int32)
    dev = torch.device("cuda", device)
    x = torch.from_numpy(np.ascontiguousarray(seq[:, :-1])).to(dev)
    y = torch.from_numpy(np.ascontiguousarray(seq[:, 1:])).to(dev)
    for _ in range(warmup):
        m.train_step_device(x.data_ptr(), y.data_ptr(), B, T, LR)
    m.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        m.train_step_device(x.data_ptr(), y.data_ptr(), B, T, LR)
    m.synchronize()
    dt = (time.perf_counter() - t0) / steps
    loss, _ = m.last_metrics()
    from composer_amd import _lib
    nk, no = C.c_int(0), C.c_int(0)
    rc = _lib.load().cmp_train_step_launches(m._h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), B, T, C.byref(nk), C.byref(no))
    m.close()
    return {"workload": "default_config.yml Transformer (8L/16H/d256), seq=1024, B=1, dropout 0.10, bf16", "steps": steps,
            "ms_per_step": 1e3 * dt, "tokens_per_s": B * T / dt, "launches": nk.value if rc == 0 else None,
            "other_graph_nodes": no.value if rc == 0 else None, "final_loss": loss,
            "model_mfma_frac": B * T / dt * flops_per_token_train(E, L, T) / 1e12 / PEAK_BF16_TFLOPS}


def dp1_child(args):
    """The product's data-parallel path on this box: a FRESH child process runs this file under torch.distributed.run with ONE rank
    (RCCL communicator, per-block buckets on the priority side stream, Adam per bucket behind its all-reduce) and its `comm` object and
    step time come back -- so the driver-timed N=1 line also says what the gradient exchange costs when nothing can overlap badly.
    A child, never a re-exec: this process has initialised the GPU."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), "--gpus", "1", "--steps", "10", "--warmup", "3", "--no-extras",
           "--no-cpu-baseline", "--no-decode", "--config", args.config, "--dropout", str(args.dropout)] + \
          (["--batch", str(args.batch)] if args.batch else [])
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode != 0 or not line:
            return {"error": "child exited %d: %s" % (p.returncode, (p.stderr or p.stdout)[-400:])}
        doc = json.loads(line[-1])
        return {"ms_per_step": doc["ms_per_step"], "value": doc["value"], "steps": doc["steps"], "comm": doc.get("comm"),
                "note": "bench.py under torch.distributed.run with one rank, own process: the RCCL path of the 8-GPU job"}
    except Exception as e:                                   # the headline line must not depend on the child
        return {"error": repr(e)[:400]}


def self_launch(args):
    """--gpus N>1 without a launcher: run N ranks of this file under torch.distributed.run as a CHILD process (this parent
    has not touched the GPU), forward their stdout/stderr, return the launcher's exit status."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--batch", type=int, default=None, help="sequences per GPU (default: the config's)")
    ap.add_argument("--roofline-kernel", type=int, default=0, help="kernel class timed live (see KERNEL_CLASSES)")
    ap.add_argument("--roofline-every", type=int, default=5, help="the live HIP-event timing of the roofline class covers every n-th "
                                                                    "step of the timed region (two events per launch cost 0.24 ms per step at n = 1: 26.60 vs 26.37 ms)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the per-class roofline table and the b32 / c4 side runs")
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--allreduce-only", action="store_true", help="time the step's gradient-exchange message pattern alone (no compute) and "
                                                                   "print its algorithm / bus bandwidth; --steps = repetitions")
    ap.add_argument("--hog", default=None, help="measurement aid: 'WGS,USEC' -- before every step, WGS workgroups spin for USEC "
                                               "microseconds on the communication stream (a stand-in for a concurrent RCCL kernel)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    under_launcher = "RANK" in os.environ and "MASTER_PORT" in os.environ
    if args.gpus > 1 and not under_launcher:
        sys.exit(self_launch(args))
    if args.gpus != world and under_launcher:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    if args.allreduce_only and not under_launcher:
        # (one rank: RCCL copies in place -- the pattern and the plumbing still run; a child, this parent has not touched the GPU)
        args.gpus = max(1, args.gpus)
        sys.exit(self_launch(args))
    json_fd = 1
    if under_launcher:
        # RCCL's diagnostics belong on stderr, the ONE JSON line on stdout: fd 1 is pointed at stderr for the whole run (RCCL and gloo
        # write to it directly), the line goes to the saved descriptor at the end
        os.environ.setdefault("NCCL_DEBUG", "WARN")
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from composer_amd.transformer import Transformer
    from composer_amd import _lib

    cf = CONFIGS[args.config]
    E, H, L, T = cf["E"], cf["H"], cf["L"], cf["T"]
    W = T
    if under_launcher:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with stdout_to_stderr():
            dist.init_process_group("gloo", rank=rank, world_size=world)     # bootstrap + timing only; gradients go over RCCL
    if not torch.cuda.is_available() or local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d needs GPU %d, but %d HIP device(s) are visible"
                         % (rank, local_rank, torch.cuda.device_count() if torch.cuda.is_available() else 0))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    Bq = args.batch or cf["B"]
    model = Transformer(V, E, W, L, H, attention_dropout_rate=args.dropout, residual_dropout_rate=args.dropout,
                        dtype="bf16", seed=1000, max_batch=Bq, max_seq=T, device=local_rank)
    model.initialize_parameters(0)                  # identical replicas; cmp_dp_init folds the rank into the dropout seed
    if under_launcher:      # every launched run (also 1 rank) takes the RCCL path: buckets, side stream, 1/N scale
        with stdout_to_stderr():
            uid = [Transformer.new_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            model.init_data_parallel(rank, world, uid[0])
            model.synchronize()

    if args.allreduce_only:
        reps = max(1, args.steps)
        model.synchronize()
        dist.barrier()
        r = model.all_reduce_pattern(reps)
        gathered = [None] * world
        dist.all_gather_object(gathered, {"rank": rank, "ms": r["ms"]})
        if rank == 0:
            ms = max(g["ms"] for g in gathered)
            alg = r["bytes"] / (ms * 1e-3) / 1e9
            out = {"metric": "gradient all-reduce of one train step alone (product bucket pattern, no compute)", "value": alg,
                   "unit": "GB/s (algorithm bandwidth: bytes of one rank's gradients / time)", "n_gpus": world, "reps": reps,
                   "ms_per_pattern": ms, "bytes": r["bytes"], "messages": r["messages"],
                   "bus_bandwidth_gbs": alg * (2.0 * (world - 1) / world if world > 1 else 1.0),
                   "bus_bandwidth_note": "ring all-reduce moves 2(N-1)/N of the message over every link; xGMI: ~153 GB/s per link "
                                         "(at N = 1 RCCL copies in place: the figure is a device-memory copy rate)",
                   "config": {"workload": "%s gradient buckets: %d messages" % (cf["label"], r["messages"])},
                   "ranks": sorted(gathered, key=lambda g: g["rank"]), "runtime": _lib.runtime_info()}
            os.write(json_fd, (json.dumps(out) + "\n").encode())
        model.close()
        dist.destroy_process_group()
        return

    # synthetic inputs, resident in HBM before the timed region
    n_data = 4
    rng = np.random.default_rng(1234)
    seq = rng.integers(0, V, size=(n_data, world * Bq, T + 1), dtype=np.int32)
    mine = seq[:, rank * Bq:(rank + 1) * Bq]
    xs = [torch.from_numpy(np.ascontiguousarray(mine[i, :, :-1])).to(dev) for i in range(n_data)]
    ys = [torch.from_numpy(np.ascontiguousarray(mine[i, :, 1:])).to(dev) for i in range(n_data)]
    torch.cuda.synchronize()

    hog = [int(v) for v in args.hog.split(",")] if args.hog else None

    def step(i):
        if hog:
            _lib.check(_lib.load().cmp_dp_test_hog(model._ctx, hog[0], hog[1]), "cmp_dp_test_hog")
        model.train_step_device(xs[i % n_data].data_ptr(), ys[i % n_data].data_ptr(), Bq, T, LR)

    def fence():
        model.synchronize()
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()

    for i in range(args.warmup):
        step(i)
    fence()
    if under_launcher:
        model.dp_stats(reset=True)
    lib = _lib.load()
    lib.cmp_prof_begin(args.roofline_kernel)
    every = max(1, args.roofline_every)
    t0 = time.perf_counter()
    for i in range(args.steps):
        if every > 1:
            (lib.cmp_prof_resume if i % every == 0 else lib.cmp_prof_pause)()
        step(args.warmup + i)
    lib.cmp_prof_resume()
    fence()
    dt = time.perf_counter() - t0
    ms, n_launch, work, abytes = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
    lib.cmp_prof_end2(C.byref(ms), C.byref(n_launch), C.byref(work), C.byref(abytes))
    loss, acc = model.last_metrics()
    comm = model.dp_stats() if under_launcher else None          # the timed steps' exposed gradient-exchange time
    # the live roofline of every kernel class of the step (outside the timed region: three more steps per class)
    classes = None
    if world == 1 and not args.no_extras:
        classes = class_table(lib, step, Bq * T, args.config)
    ranks = None
    if dist.is_initialized():
        ranks = [None] * world
        dist.all_gather_object(ranks, {"rank": rank, "ms_per_step": 1e3 * dt / args.steps,
                                       "exposed_ms": comm["exposed_ms"] if comm else None, "device": local_rank})
        ranks.sort(key=lambda r: r["rank"])
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        tokens = world * Bq * T * args.steps
        value = tokens / dt
        name, bound = KERNEL_CLASSES[args.roofline_kernel]
        if n_launch.value > 0 and ms.value > 0:
            if bound == "mfma":
                achieved = work.value / (ms.value * 1e-3) / 1e12
                tr, why = pmc_traffic_note(args.roofline_kernel, Bq * T, args.config)
                roof = {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved / PEAK_BF16_TFLOPS, "traffic": tr, "traffic_note": why}
            else:
                achieved = work.value / (ms.value * 1e-3) / 1e9
                roof = {"bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": achieved / PEAK_HBM_GBS, "traffic": None}
            roof.update({"kernel": name, "launches": int(n_launch.value),
                         "avg_launch_us": 1e3 * ms.value / n_launch.value,
                         "algorithmic_per_launch": work.value / n_launch.value,
                         "algorithmic_bytes": abytes.value / n_launch.value})
        else:
            roof = None
        fpt = flops_per_token_train(E, L, T)
        out = {
            "metric": "MIDI-event tokens/sec (train, seq=%d)" % T, "value": value, "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "%s Transformer train step, seq=%d, B=%d/GPU, dropout %.2f, Adam lr 1e-3, "
                                   "synthetic int32 event ids (vocab 390), random-init weights"
                                   % (cf["label"], T, Bq, args.dropout),
                       "global_batch": world * Bq, "seq_len": T, "parallelism": "dp%d" % world,
                       "tokens_per_step": world * Bq * T},
            "model_tflops": value * fpt / 1e12,
            "model_mfma_frac": value * fpt / 1e12 / (PEAK_BF16_TFLOPS * world),
            "final_loss": loss,
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cf)
        else:
            out["cpu_baseline"] = None
        if classes is not None:
            out["classes"] = classes
        if comm is not None:
            # exposed (non-overlapped) communication per step = how long the compute stream waited, at the end of the
            # backward pass, for the communication stream (all-reduce + Adam of the buckets still in flight); rank 0's value
            out["comm"] = {"exposed_ms": comm["exposed_ms"], "steps": comm["steps"], "bytes": comm["bytes"], "buckets": comm["buckets"],
                           "ranks": world, "note": "fp32 gradient buckets (one per decoder block + ln_f + embeddings) and a 3-float "
                           "metrics message, ncclAllReduce on a highest-priority side stream, Adam per bucket behind its all-reduce"}
            out["ranks"] = ranks
            out["runtime"] = _lib.runtime_info()
    model.close()
    if rank == 0:
        if world == 1 and not args.no_extras:
            if not (args.config == "c2" and Bq == 32):
                out["b32"] = side_config("c2", 32, local_rank, args.dropout, classes=True)
            if args.config != "c4":
                out["c4"] = side_config("c4", CONFIGS["c4"]["B"], local_rank, args.dropout)
            out["forward"] = forward_bench(local_rank)
            out["default_cfg"] = default_config_bench(local_rank)
            if not under_launcher:
                d1 = dp1_child(args)
                if "ms_per_step" in d1:
                    d1["vs_plain_step"] = d1["ms_per_step"] / out["ms_per_step"]
                out["dp1"] = d1
        if world == 1 and not args.no_decode:
            out["decode"] = decode_bench(local_rank)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
