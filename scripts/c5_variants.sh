#!/bin/bash
# Times the 10 M-triangle configuration once per library variant under ray-tracer_amd/lib_variants (tuning experiments).
for d in "" ray-tracer_amd/lib_variants/*/; do
  name=${d:-default}; lib=${d:+${d}librtx_hip.so}
  RTX_HIP_LIB=$lib timeout -k 10 200 python -u bench.py --config c5 --steps 2 --warmup 1 --cpu-baseline-seconds 0 --no-extras --spp ${SPP:-16} 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print('$name', 'ms_per_step %.1f' % j['ms_per_step'], 'extend_ms %.1f' % r['kernel_ms'], 'nodes/ray %.1f leaves/ray %.1f' % (r.get('node_records_per_ray',0), r.get('leaf_records_per_ray',0)))
" || exit 1
done
