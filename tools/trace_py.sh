#!/bin/bash
# rocprofv3 kernel trace of a python tool: trace_py.sh <lines> <tool.py> [args...]
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=/tmp/trace_py_$$
n=$1; shift
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 "$@" > /dev/null 2>&1
python3 tools/kernel_stats.py $OUT | head -$n
