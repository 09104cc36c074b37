#!/bin/bash
# FCPE back-end on the GPU box: parity tests, a secondary bench line (f0_method=fcpe) and a serial-mode kernel trace.
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/r2p
mkdir -p $OUT
python -m pytest tests/test_gpu_fcpe.py -x -q -m gpu -s > $OUT/fcpe.log 2>&1
tail -15 $OUT/fcpe.log
python bench.py --f0-method fcpe --steps 10 --warmup 3 > $OUT/bench_fcpe.json 2> $OUT/bench_fcpe.err
cut -c1-700 $OUT/bench_fcpe.json
export RVCX_SERIAL=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --f0-method fcpe --steps 5 --warmup 1 --no-roofline > $OUT/bench_trace.json 2> $OUT/trace.err
python3 tools/kernel_stats.py $OUT/trace $OUT/rocprof_fcpe_serial_kernel_stats.txt | head -48
