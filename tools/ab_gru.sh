#!/bin/bash
# A/B of the BiGRU cluster kernel forms on one box: kernel durations from a rocprofv3 kernel trace of tools/bench_gru.py
# (T = 1000, 3001, 9001, four launches each; us per step = (max - min duration) / 8001 steps), then the GRU parity test per form
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for f in ${FORMS:-0 1 2 3}; do
  OUT=/tmp/abgru_$$_$f
  RVCX_GRU_FORM=$f rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 tools/bench_gru.py > /dev/null 2>&1
  python3 - $OUT $f <<'PY'
import csv, glob, os, sys
d = []
for fn in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        if "bigru" in r["Kernel_Name"]:
            d.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
d.sort()
short, long_ = d[:4], d[-4:]
print(f"form {sys.argv[2]}: T=1000 {min(short):8.1f} us   T=9001 {min(long_):8.1f} us   per step {(min(long_) - min(short)) / 8001:.4f} us")
PY
done
if [ -z "${NOTESTS:-}" ]; then
for f in ${FORMS:-1 2 3}; do
  echo "form $f tests:"; RVCX_GRU_FORM=$f python -m pytest tests/test_gpu_rmvpe_hubert.py -q -m gpu -k "gru or rmvpe" 2>&1 | tail -1
  echo "form $f, members spread over XCDs:"; RVCX_GRU_COLOCATE=0 RVCX_GRU_FORM=$f python -m pytest tests/test_gpu_rmvpe_hubert.py -q -m gpu -k "gru" 2>&1 | tail -1
done
fi
