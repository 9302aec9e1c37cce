#!/usr/bin/env python3
"""GEMM K-sweep at M=32768: separates the fixed (prologue+epilogue) cost from the per-K main-loop cost."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.kbench as kb
import torch
kb._lib.require_gpu()
torch.zeros(1, device="cuda")
flags = int(os.environ.get("KBENCH_GEMM_FLAGS", "0"))
kb.GFLAGS = flags
for N in (512, 1536):
    for out_fp32 in (False, True):
        for K in (64, 256, 512, 1024, 2048):
            kb.gemm_case("sweep N=%d fp32out=%d" % (N, out_fp32), 0, 0, 32768, N, K, out_fp32=out_fp32)
