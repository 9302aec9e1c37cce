#!/usr/bin/env python3
"""Timing experiments on the deep-pipeline GEMM: full kernel vs no-DMA (flag 32) vs no-MFMA (flag 64)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.kbench as kb
import torch
kb._lib.require_gpu(); torch.zeros(1, device="cuda")
for nm, fl in (("full", 16), ("no DMA", 16 | 32), ("no MFMA", 16 | 64), ("no DMA no MFMA", 16 | 96), ("no epilogue", 16 | 128), ("nothing", 16 | 224)):
    kb.GFLAGS = fl
    for (n, k) in ((1536, 512), (512, 2048), (2048, 512)):
        kb.gemm_case("%-14s" % nm, 0, 0, 32768, n, k, bias=True)
