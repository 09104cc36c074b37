cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6r; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5 $tag', round(d['value'],1), d['config']['micro_batches'])" >> $O/c5.txt; }
run base X=1
run bucket64 RVCX_BUCKET_FRAMES=64
run bucket256 RVCX_BUCKET_FRAMES=256
run base X=1
run maxbatch12 RVCX_MAX_BATCH=12
run bucket256_max16 RVCX_BUCKET_FRAMES=256 RVCX_MAX_BATCH=16
cat $O/c5.txt
