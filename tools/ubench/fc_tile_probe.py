import os, sys
sys.argv=['kbench']
sys.path.insert(0,'/root/repo/tools')
os.environ.setdefault('KB_B','128')
import kbench as kb
import torch
torch.zeros(1,device='cuda')
M,E=kb.M,kb.E
for rep in range(2):
    kb.gemm_case("warm", 0, 1, M, 4*E, E, bias=True, act=1)
    kb.gemm_case("c_fc default (W^T, 256 kernel)", 0, 1, M, 4*E, E, bias=True, act=1)
    kb.gemm_case("c_fc W[K,N] p4 256x256", 0, 0, M, 4*E, E, bias=True, act=1, flags=16)
    kb.gemm_case("c_fc W[K,N] p4 128x256 x2", 0, 0, M, 4*E, E, bias=True, act=1, flags=48)
    kb.gemm_case("c_attn default", 0, 1, M, 3*E, E, bias=True)
    kb.gemm_case("c_attn W[K,N] p4 128x256 x2", 0, 0, M, 3*E, E, bias=True, flags=48)
    kb.gemm_case("dgrad gelu' default", 0, 1, M, 4*E, E, act=2)
    kb.gemm_case("dgrad gelu' p4 128x256 x2 (W^T)", 0, 1, M, 4*E, E, act=2, flags=48)
