#!/bin/bash
# round 6: one clip per step of 5 ... 45 s, the three ResBlock branches on three streams (1) or on the main stream (0); one box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r6w; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-children --clip-seconds $SEC 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$SEC s $tag', round(d['value'],1), round(d['ms_per_step'],2), round(d['stage_ms']['decoder'],2))" >> $O/streams_len.txt; }
for SEC in 5 10 15 20 30 40; do run streams3 RVCX_RESBLOCK_STREAMS=1; run streams1 RVCX_RESBLOCK_STREAMS=0; run streams3 RVCX_RESBLOCK_STREAMS=1; run streams1 RVCX_RESBLOCK_STREAMS=0; done
cat $O/streams_len.txt
