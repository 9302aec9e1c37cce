#!/usr/bin/env python3
"""Per-kernel table of a rocprofv3 --kernel-trace --stats run:  python3 tools/kstats.py <dir> [top]"""
import csv, glob, sys
d = sys.argv[1]; top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("total %.3f ms" % (tot / 1e6))
    for r in rows[:top]:
        print("%8.1f us x %5d = %8.3f ms  %5.1f%%  %s" % (float(r["AverageNs"]) / 1e3, int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6,
                                                         100 * float(r["TotalDurationNs"]) / tot, r["Name"][:150]))
