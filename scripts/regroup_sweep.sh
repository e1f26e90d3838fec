#!/bin/bash
# The headline configuration with the material-coherent variant of the fused kernel, once per threshold (lanes outside a wave's
# majority class that make the workgroup trade places; 0 = the plain kernel).  Usage: scripts/regroup_sweep.sh <threshold> ...
for m in "$@"; do
  if [ "$m" = 0 ]; then unset RTX_FUSED_REGROUP; else export RTX_FUSED_REGROUP=$m; fi
  timeout -k 10 200 python -u bench.py --config ${CFG:-c2} --steps 3 --warmup 1 --cpu-baseline-seconds 0 --no-extras 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(json.dumps({'config':'${CFG:-c2}','regroup_min_minority':$m,'ms_per_step':round(j['ms_per_step'],3),'msamples_per_s':round(j['value'],1),'kernel_ms':round(j['roofline']['kernel_ms'],3),'film_mean':j['film_mean_clamped']}))
" || exit 1
done
