#!/bin/bash
# Times the headline configuration once per library variant under ray-tracer_amd/lib_variants (tuning experiments).
for d in "" ray-tracer_amd/lib_variants/*/; do
  name=${d:-default}; lib=${d:+${d}librtx_hip.so}
  RTX_HIP_LIB=$lib timeout -k 10 200 python -u bench.py --config ${CFG:-c2} --steps 3 --warmup 1 --cpu-baseline-seconds 0 --no-extras 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$name', 'ms_per_step %.2f' % j['ms_per_step'], 'Msamples/s %.0f' % j['value'])
" || exit 1
done
