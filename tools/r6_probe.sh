#!/bin/bash
# round 6: F0 model alone with / without the weight-stationary tile, B = 1 and 16; kernel trace of the B = 1 run
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r6b; mkdir -p $O
export RVCX_DEBUG=1
for ws in 1 0; do for B in 1 16; do
  RVCX_CONV_WS=$ws timeout 300 python tools/prof_rmvpe.py $B > $O/prof_rmvpe_ws${ws}_B$B.txt 2>&1
done; done
RVCX_CONV_WS=0 timeout 300 python tools/bench_ws.py > $O/bench_ws_base.txt 2>&1
RVCX_CONV_WS=0 timeout 300 python tools/bench_ws.py 1d > $O/bench_ws_1d_base.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for ws in 1 0; do
  rm -rf /tmp/rp$ws; RVCX_CONV_WS=$ws timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/rp$ws -- python3 $OLDPWD/tools/prof_rmvpe.py 1 > /dev/null 2>&1
  python3 $OLDPWD/tools/kernel_stats.py /tmp/rp$ws $OLDPWD/$O/rocprof_rmvpe_ws${ws}_B1.txt > /dev/null
done
