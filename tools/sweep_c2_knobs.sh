#!/bin/bash
# round 6: the three ResBlock branches of an NSF stage on three streams (RVCX_RESBLOCK_STREAMS=1), on main + aux[0] (2), on the
# main stream (0) -- C2, C3, C5; repeated on one box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r6x; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --warmup 2 --no-cpu-baseline --no-roofline --no-children $ARGS 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$ARGS $tag', round(d['value'],1), {k: round(v,2) for k,v in d['stage_ms'].items() if k in ('enc_p','flow','decoder','total')})" >> $O/streams.txt; }
for ARGS in "--steps 15" "--workload c5 --steps 2" "--workload c3 --steps 2"; do
for i in 1 2; do run three RVCX_RESBLOCK_STREAMS=1; run two RVCX_RESBLOCK_STREAMS=2; run one RVCX_RESBLOCK_STREAMS=0; done; done
cat $O/streams.txt
