cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r7a; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-roofline --no-children 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2 $tag', round(d['value'],1), {k: round(v,2) for k,v in d['stage_ms'].items() if k in ('rmvpe','hubert','enc_p','flow','decoder')})" >> $O/misc.txt; }
run base X=1
run att_blocks100 RVCX_ATT_BLOCKS=100
run att_blocks400 RVCX_ATT_BLOCKS=400
run base X=1
run gru_form1 RVCX_GRU_FORM=1
run hubert_after5 RVCX_HUBERT_AFTER=5
run base X=1
run hubert_on_main RVCX_HUBERT_ON=main
run conv3thin_wgs4 RVCX_CONV3_THIN_WGS=4
run base X=1
cat $O/misc.txt
