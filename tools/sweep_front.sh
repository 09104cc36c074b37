#!/bin/bash
# C2 front-end scheduling sweep: where in the F0 model's launch sequence HuBERT is enqueued (RVCX_HUBERT_AFTER: behind U-Net
# encoder level 0..4, 5 = behind the intermediate layers, 6 = behind the whole U-Net = default), whether its stream also waits
# there on the GPU (RVCX_HUBERT_GATE, default 1), and HuBERT on the main stream instead of aux[0] (third field "main")
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for cfg in ${CFGS:-"6 1 aux" "6 0 aux" "5 1 aux" "4 1 aux" "2 1 aux" "2 0 aux" "6 1 main"}; do
  set -- $cfg
  [ "$3" = "main" ] && export RVCX_HUBERT_ON=main || unset RVCX_HUBERT_ON
  echo -n "after $1 gate $2 on $3: "; RVCX_HUBERT_AFTER=$1 RVCX_HUBERT_GATE=$2 python bench.py --no-cpu-baseline --no-children --no-roofline --steps 20 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d['stage_ms']; print(round(d['value'],1), round(d['ms_per_step'],3), {k: round(v,2) for k,v in s.items() if k in ('rmvpe','hubert','enc_p','flow','decoder')})"
done
done
