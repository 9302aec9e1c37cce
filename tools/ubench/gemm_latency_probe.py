import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ.setdefault('KB_B', '128')
import kbench as kb
import torch
kb._lib.require_gpu(); torch.zeros(1, device='cuda')
kb.GFLAGS = 0
for m in (65536, 49152, 32768, 24576, 16384, 8192, 1024):
    for k in (2048, 8192):
        us = kb.gemm_case("M=%d" % m, 0, 1, m, 512, k)
        items = (m // 256) * 2
        rounds = max(1.0, items / 256.0)
        print("   per item %.2f us, per k-step %.3f us (items %d)" % (us / rounds, us / rounds / (k // 64), items))
