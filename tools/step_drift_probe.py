import sys, os, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from composer_amd.transformer import Transformer
E,H,L,T,B=512,8,6,1024,128
m = Transformer(390, E, T, L, H, attention_dropout_rate=0.1, residual_dropout_rate=0.1, dtype="bf16", seed=1000, max_batch=B, max_seq=T)
m.initialize_parameters(0)
rng = np.random.default_rng(1234)
seq = rng.integers(0, 390, size=(2, B, T + 1), dtype=np.int32)
xs = [torch.from_numpy(np.ascontiguousarray(seq[i, :, :-1])).cuda() for i in range(2)]
ys = [torch.from_numpy(np.ascontiguousarray(seq[i, :, 1:])).cuda() for i in range(2)]
ts=[]
for i in range(60):
    t0=time.perf_counter()
    m.train_step_device(xs[i % 2].data_ptr(), ys[i % 2].data_ptr(), B, T, 1e-3)
    m.synchronize()
    ts.append(1e3*(time.perf_counter()-t0))
print(' '.join('%.2f'%t for t in ts))
