#!/usr/bin/env python3
"""Coarse per-queue activity of a `rocprofv3 --kernel-trace --output-format csv` run: for every hardware queue the
segments of continuous activity (gaps above `gap_ms` split segments) in the last `window_ms` of the trace, with the
kernel that took most of each segment.  usage: queue_segments.py <dir> [window_ms=2500] [gap_ms=1.0]"""
import collections, csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
rows.sort()
window = float(sys.argv[2]) if len(sys.argv) > 2 else 2500.0
gap = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
t_end = max(r[1] for r in rows)
t0 = t_end - int(window * 1e6)
rows = [r for r in rows if r[0] >= t0]
for q in sorted({r[2] for r in rows}):
    ks = [r for r in rows if r[2] == q]
    segs, cur = [], [ks[0]]
    for r in ks[1:]:
        if r[0] - cur[-1][1] > gap * 1e6:
            segs.append(cur)
            cur = []
        cur.append(r)
    segs.append(cur)
    print(f"queue {q}: {len(ks)} dispatches, {len(segs)} segments")
    for sg in segs:
        acc = collections.Counter()
        for r in sg:
            acc[r[3][:44]] += r[1] - r[0]
        top = acc.most_common(1)[0]
        busy = sum(r[1] - r[0] for r in sg)
        print(f"   {(sg[0][0] - t0) / 1e6:9.2f} -> {(sg[-1][1] - t0) / 1e6:9.2f} ms  ({(sg[-1][1] - sg[0][0]) / 1e6:8.2f} ms, busy {busy / 1e6:8.2f}, "
              f"{len(sg):5d} launches)  top: {top[0]} {top[1] / 1e6:.1f} ms")
