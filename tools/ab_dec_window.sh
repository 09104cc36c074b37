#!/bin/bash
# Round 6: the decoder window (RVCX_DEC_WINDOW 1 = default / 0 = every frame of every decoder call) on C2, C3, C5 and the
# 95 s clip, A/B/A/B on one box.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6w; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-roofline --no-children $BARGS 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$WL $tag', round(d['value'],1), {k: round(v,2) for k,v in (d.get('stage_ms') or {}).items() if k in ('rmvpe','hubert','enc_p','flow','decoder','total')})" >> $O/ab.txt; }
for rep in 1 2; do
  WL=c2 BARGS="--steps 20 --warmup 4"; run window1 X=1; run window0 RVCX_DEC_WINDOW=0
done
WL=c3 BARGS="--workload c3 --steps 2 --warmup 1"; run window1 X=1; run window0 RVCX_DEC_WINDOW=0; run window1 X=1
WL=c5 BARGS="--workload c5 --steps 2 --warmup 1"; run window1 X=1; run window0 RVCX_DEC_WINDOW=0; run window1 X=1
WL=c2_95s BARGS="--clip-seconds 95 --steps 5 --warmup 2"; run window1 X=1; run window0 RVCX_DEC_WINDOW=0
cat $O/ab.txt
