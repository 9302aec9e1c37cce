#!/usr/bin/env python3
"""Does the operand access pattern bound the 256x256 persistent GEMM?  Same FLOPs, A given K-major ([M,K]: every stage reads
128 B from each of 256 rows K*2 bytes apart) or contraction-slow ([K,M]: every stage reads 64 runs of 512 B), forced onto the
2-stage 256x256 kernel (CMP_GEMM_TILE256).  python tools/ubench/gemm_layout_probe.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ.setdefault('KB_B', '128')
import kbench as kb
import torch
kb._lib.require_gpu(); torch.zeros(1, device='cuda')
kb.GFLAGS = 8
for m, n, k in ((131072, 512, 2048), (65536, 512, 8192), (131072, 512, 512)):
    for ta, tb in ((0, 1), (1, 1), (0, 0), (1, 0)):
        kb.gemm_case("ta=%d tb=%d" % (ta, tb), ta, tb, m, n, k)
