#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc run:  python tools/pmc_summary.py <dir> [name-filter]"""
import csv, glob, sys, collections
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if flt in k:
            acc[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-32s n=%4d avg=%.4g" % (c, len(v), sum(v) / len(v)))
