#!/bin/bash
# Round 6 probe: HuBERT of a single clip on a CU-masked stream (aux[1], N CUs, sharing the idle main stream's hardware queue)
# beside the WHOLE F0 model, no wait behind the U-Net -- against the default (HuBERT on aux[0] behind the U-Net).  C2, one box.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6c; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-roofline --no-children 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2 $tag', round(d['value'],1), {k: round(v,2) for k,v in d['stage_ms'].items() if k in ('rmvpe','hubert','enc_p','flow','decoder','total')})" >> $O/cus.txt; }
run base X=1
for n in 64 96 128 160 192; do
  run cus$n RVCX_HUBERT_CUS=$n RVCX_HUBERT_ON=aux1 RVCX_HUBERT_GATE=0
done
run base X=1
run cus128_gate RVCX_HUBERT_CUS=128 RVCX_HUBERT_ON=aux1
run aux1_nogate_nomask RVCX_HUBERT_ON=aux1 RVCX_HUBERT_GATE=0
run cus128_after0 RVCX_HUBERT_CUS=128 RVCX_HUBERT_ON=aux1 RVCX_HUBERT_GATE=0 RVCX_HUBERT_AFTER=0
cat $O/cus.txt
