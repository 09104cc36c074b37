#!/usr/bin/env python3
"""Idle gaps of one single-clip conversion from a `rocprofv3 --kernel-trace --output-format csv` run of
`bench.py --no-children --no-cpu-baseline --no-roofline`: the last step is cut out at its first kernel (odd_ext_kernel, the
high-pass filter's extension), the union of all kernels' intervals (every queue) is formed and the moments with NO kernel
on the device are listed with the kernels either side.  usage: step_timeline.py <dir> [min_gap_us=5]"""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
rows.sort()
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
starts = [i for i, r in enumerate(rows) if "odd_ext_kernel" in r[3]]
a, b = starts[-2], starts[-1]          # the step before the last (the last one may be the cpu-side tail)
step = rows[a:b]
t0 = step[0][0]
print(f"step: {len(step)} dispatches, {(max(r[1] for r in step) - t0) / 1e3:.1f} us from first start to last end; "
      f"to the next step's first kernel {(rows[b][0] - t0) / 1e3:.1f} us")
busy_end, idle, gaps = step[0][1], 0, []
last = step[0]
for r in step[1:]:
    if r[0] > busy_end:
        g = (r[0] - busy_end) / 1e3
        idle += g
        if g >= min_gap:
            gaps.append((g, (busy_end - t0) / 1e3, last[3][:60], r[3][:60]))
    if r[1] > busy_end:
        busy_end, last = r[1], r
print(f"device idle inside the step: {idle:.1f} us in total; gaps >= {min_gap} us:")
for g, at, ka, kb in sorted(gaps, reverse=True)[:40]:
    print(f"  {g:7.1f} us at {at:9.1f}  after {ka}  before {kb}")
