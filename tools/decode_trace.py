#!/usr/bin/env python3
"""Per-kernel duration and the gap in front of each kernel of the per-token decode chain, from a rocprofv3 --kernel-trace CSV:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/x -o k -- python3 tools/decode_bench.py
    python tools/decode_trace.py gpurun_out/x"""
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
dec = [(a, b, n) for a, b, n in rows if n.startswith(("void dec_", "dec_"))]
print("decode kernels:", len(dec))
dur, gap, cnt = collections.Counter(), collections.Counter(), collections.Counter()
prev_end = None
for a, b, n in dec[len(dec) // 2:]:
    key = n.split("(")[0][:60]
    dur[key] += b - a
    cnt[key] += 1
    if prev_end is not None and a - prev_end < 50000:
        gap[key] += a - prev_end
    prev_end = b
tot = 0.0
ntok = max(1, cnt[[k for k in cnt if "sample" in k][0]]) if any("sample" in k for k in cnt) else 1
for k in cnt:
    print("%-62s n/token %5.1f  dur %6.2f us  gap before %6.2f us" % (k, cnt[k] / ntok, dur[k] / cnt[k] / 1e3, gap[k] / cnt[k] / 1e3))
    tot += (dur[k] + gap[k]) / 1e3
print("sum per token: %.1f us over %d tokens" % (tot / ntok, ntok))
