#!/bin/bash
# C2 / C3 / C5 under different length-class widths (RVCX_BUCKET_FRAMES) and micro-batch caps (RVCX_MAX_BATCH)
mkdir -p gpurun_out
for cfg in ${CFGS:-"64 8" "128 8" "128 16" "256 8" "256 16"}; do
  set -- $cfg
  for wl in ${WLS:-c2 c3 c5}; do
    RVCX_BUCKET_FRAMES=$1 RVCX_MAX_BATCH=$2 python bench.py --workload $wl --no-cpu-baseline --no-children --no-roofline \
      > gpurun_out/sw_${wl}_bf$1_mb$2.json 2> gpurun_out/sw_${wl}_bf$1_mb$2.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/sw_${wl}_bf$1_mb$2.json").read().strip().splitlines()[-1])
print("$wl bucket_frames", $1, "max_batch", $2, "rtf %.1f" % d["value"], "ms %.1f" % d["ms_per_step"], {k: round(v, 1) for k,v in d["stage_ms"].items()})
PY
  done
done
