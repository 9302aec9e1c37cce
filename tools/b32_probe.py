#!/usr/bin/env python3
"""bench.py's B=32 side run (side_config) repeated in one process: does its 7 ms step read the same every time?   python tools/b32_probe.py"""
import sys, json
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
for i in range(3):
    r = bench.side_config("c2", 32, 0, 0.1)
    print("side_config b32: %.3f ms/step" % r["ms_per_step"], flush=True)
r = bench.side_config("c2", 32, 0, 0.1, steps=30, warmup=5)
print("side_config b32 (30 steps): %.3f ms/step" % r["ms_per_step"], flush=True)
