#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc passes (--output-format csv).  usage: pmc_summary.py <dir> [kernel substring]"""
import collections, csv, glob, os, sys
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        a = acc[r["Kernel_Name"]][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for k in sorted(acc):
    if flt and flt not in k:
        continue
    print(k[:150])
    for c, (n, v) in sorted(acc[k].items()):
        print(f"   {c:34s} launches {n:4d}  mean {v / n:16.1f}")
