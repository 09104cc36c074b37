#!/usr/bin/env python3
"""Per-CU occupancy statistics from a conv workgroup trace (RVCX_ABLATION build, RVCX_TRACE=file):
columns idx, HW_ID, XCC_ID, t_start, t_stage0, t_mainloop_end, t_end (100 MHz ticks)."""
import sys, collections
import numpy as np
d = np.loadtxt(sys.argv[1], delimiter=",", dtype=np.int64)
hw, xcc = d[:, 1], d[:, 2] & 15
cu = (hw >> 8) & 15
se = (hw >> 13) & 7
key = xcc * 1000 + se * 16 + cu
t0, t1, t2, t3 = [d[:, i].astype(np.float64) / 100.0 for i in (3, 4, 5, 6)]   # us
T0 = t0.min()
t0, t1, t2, t3 = t0 - T0, t1 - T0, t2 - T0, t3 - T0
span = t3.max()
cyc = (d[:, 2] >> 8).astype(np.float64)
print(f"shader clock while the blocks ran: {np.sum(cyc) / np.sum(t3 - t0):.0f} MHz (s_memtime cycles / s_memrealtime us, mean over blocks)")
print(f"blocks {len(d)}  CUs seen {len(set(key))}  kernel span {span:.1f} us")
print(f"block dur us: mean {np.mean(t3-t0):.1f} p5 {np.percentile(t3-t0,5):.1f} p95 {np.percentile(t3-t0,95):.1f} | prologue {np.mean(t1-t0):.1f} mainloop {np.mean(t2-t1):.1f} epilogue {np.mean(t3-t2):.1f}")
per = collections.defaultdict(list)
for k, a, b in zip(key, t0, t3):
    per[k].append((a, b))
nb = np.array([len(v) for v in per.values()])
print(f"blocks per CU: min {nb.min()} mean {nb.mean():.2f} max {nb.max()}")
ends = np.array([max(b for a, b in v) for v in per.values()])
firsts = np.array([min(a for a, b in v) for v in per.values()])
print(f"CU finish time: min {ends.min():.1f} mean {ends.mean():.1f} max {ends.max():.1f} us; first start max {firsts.max():.1f}")
# time-averaged resident blocks per CU
res = sum(b - a for v in per.values() for a, b in v) / (len(per) * span)
print(f"time-avg resident blocks per CU over the span: {res:.2f}")
# lockstep measure: fraction of (CU, time) where >= half of the CU's resident blocks are in prologue/epilogue
grid = np.arange(0, span, 0.5)
ph = collections.defaultdict(lambda: [np.zeros(len(grid)), np.zeros(len(grid))])
for k, a, b, c, e in zip(key, t0, t1, t2, t3):
    r, m = ph[k]
    r += (grid >= a) & (grid < e)
    m += (grid >= b) & (grid < c)
nomf = np.mean([np.mean((r > 0) & (m == 0)) for r, m in ph.values()])
idle = np.mean([np.mean(r == 0) for r, m in ph.values()])
one = np.mean([np.mean((m == 1)) for r, m in ph.values()])
print(f"CU-time with no resident block {idle:.3f}; resident but none in main loop {nomf:.3f}; exactly one in main loop {one:.3f}")
