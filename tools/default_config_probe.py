import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from composer_amd.transformer import Transformer
V=390
for (E,H,L,T,B) in ((256,16,8,1024,1),(256,16,8,1024,8),(256,16,8,256,1)):
    m = Transformer(V, E, T, L, H, dtype="bf16", seed=0, max_batch=B, max_seq=T)
    rng=np.random.default_rng(0)
    x=rng.integers(0,V,(B,T),dtype=np.int32); y=rng.integers(0,V,(B,T),dtype=np.int32)
    for _ in range(5): m.train_step(x,y,1e-3)
    t0=time.perf_counter(); n=50
    tk=[]
    for i in range(n):
        tk.append(m.train_step_async(x,y,1e-3))
        if len(tk)>1: m.step_metrics(tk.pop(0))
    m.step_metrics(tk.pop(0)); m.synchronize()
    dt=(time.perf_counter()-t0)/n
    # device-only time: submit many then sync
    xd=torch.from_numpy(x).cuda(); yd=torch.from_numpy(y).cuda()
    m.synchronize(); t0=time.perf_counter()
    for i in range(n): m.train_step_device(xd.data_ptr(), yd.data_ptr(), B, T, 1e-3)
    m.synchronize(); dd=(time.perf_counter()-t0)/n
    print("E=%d L=%d T=%d B=%d: pipelined loop %.2f ms/step (%.1f k tok/s); device-pointer steps back to back %.2f ms/step" % (E,L,T,B,dt*1e3,B*T/dt/1e3,dd*1e3))
    m.close()
