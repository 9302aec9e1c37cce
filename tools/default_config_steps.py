import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from composer_amd.transformer import Transformer
V,E,H,L,T,B=390,256,16,8,1024,1
m = Transformer(V, E, T, L, H, dtype="bf16", seed=0, max_batch=B, max_seq=T)
rng=np.random.default_rng(0)
x=rng.integers(0,V,(B,T),dtype=np.int32); y=rng.integers(0,V,(B,T),dtype=np.int32)
for _ in range(30): m.train_step(x,y,1e-3)
m.close()
