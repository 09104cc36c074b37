cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
for n in ${CUS:-184 200 216 232 248 256}; do
  echo -n "HUBERT_CUS $n: "; RVCX_HUBERT_CUS=$n python bench.py --no-cpu-baseline --no-children --no-roofline --steps 20 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d['stage_ms']; print(round(d['value'],1), round(d['ms_per_step'],3), 'rmvpe', round(s['rmvpe'],2), 'hubert', round(s['hubert'],2))"
done
done
