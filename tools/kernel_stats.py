#!/usr/bin/env python3
"""Per-kernel table (calls, total / average / min / max duration, share) from a `rocprofv3 --kernel-trace
--output-format csv` run.  usage: kernel_stats.py <dir> [out.txt]"""
import collections, csv, glob, os, sys
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = sorted(((k, len(v), sum(v), sum(v) / len(v), min(v), max(v)) for k, v in acc.items()), key=lambda r: -r[2])
tot = sum(r[2] for r in rows) or 1
lines = [f"{'kernel':110s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}"]
for r in rows[:45]:
    lines.append(f"{r[0][:110]:110s} {r[1]:7d} {r[2] / 1e6:10.3f} {r[3] / 1e3:10.2f} {r[4] / 1e3:9.2f} {r[5] / 1e3:9.2f} "
                 f"{100 * r[2] / tot:6.2f}")
lines.append(f"TOTAL kernel time {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
txt = "\n".join(lines)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
