#!/bin/bash
# rocprofv3 kernel trace of the C2 bench (concurrent mode): per-queue timeline of the last step + key events of the front
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=/tmp/trace_c2_$$
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --no-cpu-baseline --no-children --no-roofline --steps 4 --warmup 2 > /dev/null 2>&1
python3 tools/timeline.py $OUT | head -40
python3 - $OUT <<'PY'
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if "iir_slice" in r[3] and (i == 0 or "iir_slice" not in rows[i - 1][3])]
starts = [s for j, s in enumerate(starts) if j == 0 or rows[s][0] - rows[starts[j - 1]][0] > 5_000_000]
step = rows[starts[-1]:]
t0 = step[0][0]
def first(pat): return next(((r[0] - t0) / 1e6, (r[1] - t0) / 1e6) for r in step if pat in r[3])
def last(pat): return [((r[0] - t0) / 1e6, (r[1] - t0) / 1e6) for r in step if pat in r[3]][-1]
for name, pat in (("mel/STFT first conv", "conv_cin1"), ("BiGRU", "bigru"), ("groupnorm (HuBERT extractor layer 0)", "groupnorm_gelu"),
                  ("last HuBERT gemm", "gemm_h3"), ("first attention", "attn_h3"), ("decode_f0", "decode_f0"), ("first resblock_pair", "resblock_pair")):
    try:
        print(f"{name:40s} first {first(pat)[0]:7.3f} - {first(pat)[1]:7.3f}   last {last(pat)[0]:7.3f} - {last(pat)[1]:7.3f} ms")
    except StopIteration:
        pass
PY
python3 - $OUT <<'PY'
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"], r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?"), r.get("LDS_Block_Size", "?"), r.get("VGPR_Count", "?")))
rows.sort()
starts = [i for i, r in enumerate(rows) if "iir_slice" in r[3] and (i == 0 or "iir_slice" not in rows[i - 1][3])]
starts = [s for j, s in enumerate(starts) if j == 0 or rows[s][0] - rows[starts[j - 1]][0] > 5_000_000]
step = rows[starts[-1]:]
t0 = step[0][0]
print("all queues, 1.0 .. 2.0 ms and 6.2 .. 6.6 ms of the last step (start, end, queue, grid, wg, lds, vgpr, kernel):")
for r in step:
    a = (r[0] - t0) / 1e6
    if 1.0 <= a <= 2.0 or 6.2 <= a <= 6.6:
        print(f"  {a:7.3f} {(r[1] - t0) / 1e6:7.3f} q{r[2]} {r[4]:>8s} {r[5]:>4s} {r[6]:>6s} {r[7]:>4s} {r[3][:90]}")
PY
