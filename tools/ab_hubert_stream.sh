#!/bin/bash
# HuBERT's stream: aux[0] (default) against aux[1] and the main stream, C2 / C3 / C5 on one box
for w in c2 c3 c5 c3 c5; do
  for on in aux0 aux1 main; do
    RVCX_HUBERT_ON=$on python bench.py --workload $w --no-children --no-roofline 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('HUBERT_ON=$on $w', round(r['value'],1), round(r['ms_per_step'],3))
"
  done
done
