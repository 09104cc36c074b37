#!/usr/bin/env python3
"""Per-stream timeline summary of one bench step from a rocprofv3 rocpd .db (kernel trace)."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
step = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = db.execute("select name,start,end,stream_id from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if 'iir_slice' in r[0]][::2]
w = rows[starts[step]:starts[step + 1]] if step + 1 < len(starts) else rows[starts[step]:]
t0 = w[0][1]
ms = lambda t: (t - t0) / 1e6
by = collections.defaultdict(list)
for r in w:
    by[r[3]].append(r)
for sid, l in sorted(by.items()):
    print(f"stream {sid}: n={len(l)} span {ms(l[0][1]):.2f}..{ms(l[-1][2]):.2f} busy {sum(r[2]-r[1] for r in l)/1e6:.2f}")
marks = ['bigru_cluster', 'groupnorm_gelu', 'embed_pitch', 'sine_prefix', 'gru_input', 'to_int16', 'decode_f0', 'attn_kernel<2>']
seen = set()
for r in w:
    for m in marks:
        if m in r[0] and m not in seen:
            seen.add(m)
            print(f"  first {m:18s} stream {r[3]} {ms(r[1]):.2f}..{ms(r[2]):.2f}")
last_attn2 = [r for r in w if 'attn_kernel<2>' in r[0]]
if last_attn2:
    print(f"  last attn_kernel<2> ends {ms(last_attn2[-1][2]):.2f}")
print(f"step total {ms(w[-1][2]):.2f} ms")
