cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6z; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-roofline --no-children $ARGS 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$ARGS $tag', round(d['value'],1), {k: round(v,2) for k,v in d['stage_ms'].items() if k in ('enc_p','flow','decoder')})" >> $O/hwq.txt; }
ARGS=""
run two X=1
run three RVCX_RESBLOCK_STREAMS=1
run three_hwq8 RVCX_RESBLOCK_STREAMS=1 GPU_MAX_HW_QUEUES=8
run two_hwq8 GPU_MAX_HW_QUEUES=8
run two X=1
ARGS="--workload c5 --steps 2"
run two X=1
run three_hwq8 RVCX_RESBLOCK_STREAMS=1 GPU_MAX_HW_QUEUES=8
run three RVCX_RESBLOCK_STREAMS=1
run two_hwq8 GPU_MAX_HW_QUEUES=8
cat $O/hwq.txt
