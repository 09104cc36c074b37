cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-exact-fp32 --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['stage_ms'].items() if k in ('rmvpe','hubert','enc_p','flow','decoder')})"; }
run base
RVCX_RESBLOCK_STREAMS=0 run rbs0
RVCX_HUBERT_CUS=216 run hub216
RVCX_HUBERT_CUS=240 run hub240
RVCX_HUBERT_CUS=0 run hub0
run base2
RVCX_F0_PRIORITY=0 run f0prio0
