#!/usr/bin/env python3
"""Live roofline of EVERY kernel class of the train step (bench.py --roofline-kernel 0..7, HIP events on the launch stream inside the
timed region), one short bench run per class:   python tools/roofline_table.py [c2|c4] > profiles/<tag>_roofline_table.txt"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
print("%-96s %8s %9s %10s %7s" % ("kernel class (bench.py --config %s --roofline-kernel n)" % cfg, "launches", "avg us", "achieved", "frac"))
for cls in range(8):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--roofline-kernel", str(cls), "--no-cpu-baseline",
                        "--no-decode", "--no-extras", "--steps", "10", "--warmup", "3"], capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().split("\n")[-1])
        rf = d["roofline"]
        print("%-96s %8d %9.1f %7.1f %s %6.1f%%   (step %.2f ms)" % ("%d %s" % (cls, rf["kernel"][:92]), rf["launches"], rf["avg_launch_us"],
              rf["achieved"], rf["unit"], 100 * rf["frac"], d["ms_per_step"]))
    except Exception as e:
        print(cls, "failed", e, r.stderr[-300:])
