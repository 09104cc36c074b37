#!/bin/bash
# C2 front-end scheduling sweep: HuBERT on its own stream (RVCX_HUBERT_ON_AUX unset) or on the decoder's aux stream 0/1, where in the
# F0 model's launch sequence it is enqueued (RVCX_HUBERT_AFTER: behind U-Net encoder level 0..4, 5 = behind the intermediate
# layers, 6 = behind the whole U-Net) and whether its stream also waits there on the GPU (RVCX_HUBERT_GATE)
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for cfg in ${CFGS:-"- 2 0" "0 2 0" "0 2 1" "0 3 1" "0 4 1" "0 5 1" "0 6 1" "1 5 1"}; do
  set -- $cfg
  [ "$1" = "-" ] && unset RVCX_HUBERT_ON_AUX || export RVCX_HUBERT_ON_AUX=$1
  echo -n "aux $1 after $2 gate $3: "; RVCX_HUBERT_AFTER=$2 RVCX_HUBERT_GATE=$3 python bench.py --no-cpu-baseline --no-children --no-roofline --steps 20 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d['stage_ms']; print(round(d['value'],1), round(d['ms_per_step'],3), {k: round(v,2) for k,v in s.items() if k in ('rmvpe','hubert','enc_p','flow','decoder')})"
done
done
