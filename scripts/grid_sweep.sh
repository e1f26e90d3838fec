#!/bin/bash
# Fused launches with RTX_ITEMS_PER_LANE work items per lane (0 = the resident grid): ms per step of a bench configuration for each value.
# Usage: scripts/grid_sweep.sh <config> <items> [<items> ...]
cfg=$1; shift
for m in "$@"; do
  RTX_ITEMS_PER_LANE=$m timeout -k 10 200 python -u bench.py --config $cfg --steps 3 --warmup 1 --cpu-baseline-seconds 0 --no-extras 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(json.dumps({'config':'$cfg','items_per_lane':$m,'ms_per_step':round(j['ms_per_step'],3),'msamples_per_s':round(j['value'],1),'kernel_ms':round(j['roofline']['kernel_ms'],3)}))
" || exit 1
done
