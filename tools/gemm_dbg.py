#!/usr/bin/env python3
"""Timing experiment: GEMM with the epilogue's global stores skipped (flag 64) vs the full kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.kbench as kb
import torch
kb._lib.require_gpu(); torch.zeros(1, device="cuda")
for nm, fl in (("full", 0), ("no stores", 64)):
    kb.GFLAGS = fl
    for (n, k) in ((1536, 512), (512, 512), (2048, 512), (512, 2048)):
        kb.gemm_case("%-10s" % nm, 0, 0, 32768, n, k, bias=True)
    kb.gemm_case("%-10s gelu+aux" % nm, 0, 0, 32768, 2048, 512, bias=True, act=1)
