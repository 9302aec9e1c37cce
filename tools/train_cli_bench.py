#!/usr/bin/env python3
"""Throughput of the COMMAND-LINE training path: `python -m composer_amd train transformer <dir> -c <cfg>` on a synthetic
`.data` dataset at the bench.py workload (6L/8H/d512, window 1024, B=128, bf16, dropout 0.1), next to bench.py's number.
Two runs with different --max-steps; the difference removes start-up (import, model creation, dataset load):
    tokens/s = (N2 - N1) * B * T / (t2 - t1)."""
import os, sys, time, subprocess, shutil, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, yaml
from composer_amd import cli, dataset as D

B, T = int(os.environ.get("TB_B", "128")), 1024
N1, N2 = 40, 440           # 400 steps of difference: a 0.1 s start-up jitter is 0.25 ms/step
tmp = tempfile.mkdtemp(prefix="cli_bench_")
try:
    os.makedirs(os.path.join(tmp, "data", "train"))
    first = os.path.join(tmp, "data", "train", "a0.data")
    D.write_synthetic_data_file(first, 1_000_000, seed=1)
    need = (N2 + 2) * B * (T + 1)
    for i in range(1, -(-need // 1_000_000)):
        shutil.copy(first, os.path.join(tmp, "data", "train", "a%d.data" % i))
    cfg = yaml.safe_load(open(cli.get_default_config()))
    cfg["transformer"]["model"].update(window_size=T, embedding_size=512, decoder_layers_count=6, attention_head_count=8)
    cfg["transformer"]["train"]["batch_size"] = B
    cfg["transformer"]["runtime"] = {"dtype": "bf16", "seed": 0}
    cfgp = os.path.join(tmp, "cfg.yml")
    yaml.safe_dump(cfg, open(cfgp, "w"))
    times = {}
    for n in (N1, N2):
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, "-m", "composer_amd", "train", "transformer", os.path.join(tmp, "data"), "-c", cfgp, "-e", "2",
                            "--logdir", os.path.join(tmp, "logs%d" % n), "--max-steps", str(n), "--no-show-progress-bar",
                            "--save-freq", "1000000"], cwd=ROOT, capture_output=True, text=True)
        times[n] = time.perf_counter() - t0
        if r.returncode != 0:
            print(r.stdout[-2000:], r.stderr[-2000:]); sys.exit(1)
    dt = times[N2] - times[N1]
    print("composer train (CLI): %d steps in %.2f s, %d steps in %.2f s -> %.1f ms/step, %.3f M tokens/s at B=%d"
          % (N1, times[N1], N2, times[N2], 1e3 * dt / (N2 - N1), (N2 - N1) * B * T / dt / 1e6, B))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
