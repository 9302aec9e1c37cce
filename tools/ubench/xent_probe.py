"""softmax cross-entropy kernel on the benchmark logits shape ([tokens, 448] fp32 in, bf16 gradient out): us and GB/s.
    COMPOSER_HIP_LIB=composer_amd/lib/<variant>.so python tools/ubench/xent_probe.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.getcwd())
from composer_amd import _lib
lib=_lib.load()
P=lambda t: C.c_void_p(t.data_ptr())
st=lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
for rows in (131072, 32768):
    V, ldz = 390, 448
    z=torch.randn(rows, ldz, device="cuda"); y=torch.randint(0,V,(rows,),device="cuda",dtype=torch.int32)
    dz=torch.zeros(rows, ldz, device="cuda", dtype=torch.bfloat16); rl=torch.zeros(rows,device="cuda"); rc=torch.zeros(rows,device="cuda",dtype=torch.int32)
    f=lambda: lib.cmp_k_softmax_xent(st(), P(z), ldz, P(y), P(dz), P(rl), P(rc), rows, V, 1.0/rows, 1)
    for _ in range(20): f()
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): f()
    b.record(); torch.cuda.synchronize()
    us=a.elapsed_time(b)*1e3/50
    print(rows, "%.1f us  %.0f GB/s" % (us, rows*(ldz*4+ldz*2)/us/1e3))
