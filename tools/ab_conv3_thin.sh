for rep in 1 2; do
for m in 0 1; do
  RVCX_CONV3_THIN=$m python bench.py --no-children --no-roofline 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('THIN=$m C2', round(r['value'],1), round(r['ms_per_step'],3), {k:round(v,2) for k,v in r['stage_ms'].items() if k in ('rmvpe','hubert','decoder','total')})
"
done
done
for m in 0 1; do
  RVCX_CONV3_THIN=$m python bench.py --workload c3 --no-children --no-roofline 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('THIN=$m C3', round(r['value'],1), round(r['ms_per_step'],2))
"
done
