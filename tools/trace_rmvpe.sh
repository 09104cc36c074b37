#!/bin/bash
# kernel trace of the F0 model alone (tools/prof_rmvpe.py B seconds): per-kernel table + the launch-to-launch timeline of one call
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=/tmp/trace_rmvpe_$$
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 tools/prof_rmvpe.py ${1:-1} ${2:-30} > /dev/null 2>&1
python3 tools/kernel_stats.py $OUT | head -${3:-32}
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the last call of the process = the last run of kernels starting with reflect_pad
starts = [i for i, r in enumerate(rows) if "reflect_pad" in r[2]]
i0 = starts[-1]
call = rows[i0:]
span = (call[-1][1] - call[0][0]) / 1e3
busy = sum(e - s for s, e, _ in call) / 1e3
gaps = sum(max(0, call[i + 1][0] - call[i][1]) for i in range(len(call) - 1)) / 1e3
print(f"last call: {len(call)} kernels, span {span:.1f} us, sum of kernel times {busy:.1f} us, idle gaps {gaps:.1f} us")
PY
