#!/bin/bash
# A/B/A/B of one environment switch in bench.py on one box: ab_env.sh VAR [workloads...]   (default: c2 c2 c3 c5)
VAR=$1; shift
WL=${@:-"c2 c2 c3 c5"}
for w in $WL; do
  for m in 0 1; do
    env $VAR=$m python bench.py --workload $w --no-children --no-roofline 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
st=r.get('stage_ms') or {}
print('$VAR=$m $w', round(r['value'],1), round(r['ms_per_step'],3), {k:round(v,2) for k,v in st.items() if k in ('rmvpe','hubert','decoder','total')})
"
  done
done
