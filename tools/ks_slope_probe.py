import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from composer_amd.transformer import Transformer
V, E, H, L = 390, 256, 16, 2
T = int(sys.argv[1]); p = float(sys.argv[2])
m = Transformer(V, E, T, L, H, attention_dropout_rate=p, residual_dropout_rate=p, dtype="bf16", seed=0, max_batch=1, max_seq=T)
rng = np.random.default_rng(0)
x = rng.integers(0, V, (1, T), dtype=np.int32); y = rng.integers(0, V, (1, T), dtype=np.int32)
for _ in range(40): m.train_step(x, y, 1e-3)
m.close()
