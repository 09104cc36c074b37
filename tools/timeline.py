#!/usr/bin/env python3
"""Per-queue timeline of the LAST conversion step in a `rocprofv3 --kernel-trace --output-format csv` run of bench.py
(concurrent mode): for each hardware queue the span, the busy time, the idle time inside the span and the dispatch
count; chip-wide idle time (no kernel running on any queue); the largest gaps per queue with the kernels around them.
usage: timeline.py <dir> [out.txt]"""
import csv, glob, os, sys

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
rows.sort()
# a step starts with the float64 high-pass: the first iir_slice_kernel after a pause
starts = [i for i, r in enumerate(rows) if "iir_slice" in r[3] and (i == 0 or "iir_slice" not in rows[i - 1][3])]
starts = [s for j, s in enumerate(starts) if j == 0 or rows[s][0] - rows[starts[j - 1]][0] > 5_000_000]
lo = starts[-1]
step = rows[lo:]
t0 = step[0][0]
out = [f"last step: {len(step)} dispatches, {(max(r[1] for r in step) - t0) / 1e6:.3f} ms from the first high-pass kernel to the last kernel end"]
queues = sorted({r[2] for r in step})
for q in queues:
    ks = [r for r in step if r[2] == q]
    busy = sum(r[1] - r[0] for r in ks)
    span = ks[-1][1] - ks[0][0]
    out.append(f"queue {q}: {len(ks):4d} dispatches, starts at {(ks[0][0] - t0) / 1e6:7.3f} ms, span {span / 1e6:7.3f} ms, busy {busy / 1e6:7.3f} ms, "
               f"idle inside span {(span - busy) / 1e6:7.3f} ms  first={ks[0][3][:40]} last={ks[-1][3][:40]}")
    gaps = sorted(((ks[i + 1][0] - ks[i][1], i) for i in range(len(ks) - 1)), reverse=True)[:6]
    for g, i in gaps:
        out.append(f"      gap {g / 1e3:8.1f} us at {(ks[i][1] - t0) / 1e6:7.3f} ms after {ks[i][3][:48]:48s} before {ks[i + 1][3][:48]}")
# chip-wide idle
ev = sorted([(r[0], 1) for r in step] + [(r[1], -1) for r in step])
depth, idle, last = 0, 0, None
for t, d in ev:
    if depth == 0 and last is not None:
        idle += t - last
    depth += d
    if depth == 0:
        last = t
out.append(f"chip-wide idle inside the step: {idle / 1e6:.3f} ms")
txt = "\n".join(out)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
