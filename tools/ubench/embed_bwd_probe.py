import ctypes as C, os, sys
import torch
sys.path.insert(0, '/root/repo')
from composer_amd import _lib
lib=_lib.load()
P=lambda t: C.c_void_p(t.data_ptr())
st=lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
V,E,T=390,512,1024
for B in (128,32):
    ids=torch.randint(0,V,(B,T),dtype=torch.int32).cuda()
    dh=torch.randn(B*T,E,device='cuda').to(torch.bfloat16)
    dwte=torch.zeros(V,E,device='cuda'); dwpe=torch.zeros(T,E,device='cuda')
    for i in range(6):
        lib.cmp_k_embed_bwd_v(st(),P(ids),P(dh),P(dwte),P(dwpe),B,T,E,0,1,float(os.environ.get('EMB_P','0.1')),5,2,V)
        lib.cmp_k_embed_bwd(st(),P(ids),P(dh),P(dwte),P(dwpe),B,T,E,0,1,0.1,5,2)
    torch.cuda.synchronize()
