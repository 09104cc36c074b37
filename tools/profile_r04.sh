#!/bin/bash
# Round-4 kernel traces (run on the GPU box through gpurun): rocprofv3 --kernel-trace of a bench workload in SERIAL
# mode (every launch on the library's one stream: a kernel's duration is its own).  usage: profile_r04.sh <tag> <bench args...>
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp RVCX_SERIAL=1
TAG=$1; shift
OUT=gpurun_out/prof_r04_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline --no-children --no-roofline "$@" \
  > $OUT/bench.json 2> $OUT/trace.err
python3 tools/kernel_stats.py $OUT/trace $OUT/rocprof_r04_${TAG}_kernel_stats.txt > /dev/null
rm -rf $OUT/trace
head -50 $OUT/rocprof_r04_${TAG}_kernel_stats.txt
tail -c 600 $OUT/bench.json
