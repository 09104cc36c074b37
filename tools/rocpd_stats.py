#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) kernel trace into the per-kernel stats table that
`rocprofv3 --kernel-trace --stats` prints: calls, total / average / min / max duration, share."""
import sqlite3
import sys


def main(path, out=None, top=40):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = db.execute(f"select {name}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                      f"from kernels group by {name} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = [f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}"]
    for r in rows[:top]:
        lines.append(f"{r[0][:90]:90s} {r[1]:7d} {r[2] / 1e6:10.3f} {r[3] / 1e3:10.2f} {r[4] / 1e3:9.2f} "
                     f"{r[5] / 1e3:9.2f} {100 * r[2] / tot:6.2f}")
    lines.append(f"TOTAL kernel time {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
