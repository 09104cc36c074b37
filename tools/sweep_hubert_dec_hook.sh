#!/bin/bash
# Round 6: HuBERT enqueued (and gated) behind decoder level d of the F0 U-Net instead of behind the whole U-Net
# (RVCX_HUBERT_AFTER_DEC; the last decoder levels are the big bandwidth-bound maps).  C2, A/B/A/B on one box.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6h; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-roofline --no-children 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2 $tag', round(d['value'],1), {k: round(v,2) for k,v in d['stage_ms'].items() if k in ('highpass','rmvpe','hubert','enc_p','flow','decoder','total')})" >> $O/dec_hook.txt; }
for rep in 1 2; do
  run base X=1
  run dec3 RVCX_HUBERT_AFTER_DEC=3
  run dec2 RVCX_HUBERT_AFTER_DEC=2
  run dec1 RVCX_HUBERT_AFTER_DEC=1
done
run base X=1
cat $O/dec_hook.txt
