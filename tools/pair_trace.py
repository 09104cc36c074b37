#!/usr/bin/env python3
"""Phase timeline of the fused ResBlock step from the per-workgroup stamps rvcx_bench_resblock_pair writes under
RVCX_PAIR_TRACE=<csv> (resblock.hip: stamp()).  usage: pair_trace.py <csv>"""
import sys
import numpy as np
blocks, cur, hdr = [], [], None
for line in open(sys.argv[1]):
    if line.startswith("#"):
        if cur:
            blocks.append((hdr, np.array(cur, dtype=np.int64)))
        hdr, cur = line[1:].strip(), []
    else:
        cur.append([int(v) for v in line.split(",")])
if cur:
    blocks.append((hdr, np.array(cur, dtype=np.int64)))
names = ["launch->first stage (loads arrive)", "c1 loop", "c1 epilogue (Y1)", "c2 loop", "epilogue (stores)"]
for hdr, a in blocks:
    t0 = a[:, 0].min()
    span = (a[:, 5].max() - t0) / 100.0
    d = np.diff(a[:, :6], axis=1) / 100.0           # us (100 MHz ticks)
    print(f"{hdr}: {len(a)} workgroups, kernel span {span:.1f} us, workgroup life mean {d.sum(1).mean():.1f} us "
          f"(min {d.sum(1).min():.1f}, max {d.sum(1).max():.1f})")
    for k, n in enumerate(names):
        print(f"   {n:36s} mean {d[:, k].mean():7.2f} us  p10 {np.percentile(d[:, k], 10):7.2f}  p90 {np.percentile(d[:, k], 90):7.2f}"
              f"  share {100 * d[:, k].mean() / d.sum(1).mean():5.1f} %")
    # rounds: workgroups sorted by start time, occupancy over time
    starts = np.sort(a[:, 0] - t0) / 100.0
    ends = np.sort(a[:, 5] - t0) / 100.0
    print(f"   first 256 start within {starts[min(255, len(starts) - 1)]:.1f} us; last start {starts[-1]:.1f} us; last end {ends[-1]:.1f} us;"
          f" CU-time used / (256 CUs x span) = {d.sum(1).sum() / (256 * span) * 100:.1f} %")
