#!/bin/bash
set -e
for v in "" ph_heap ph_skip ph_heap4 ph_arr4; do
  if [ -n "$v" ]; then export RTX_HIP_LIB=$PWD/ray-tracer_amd/lib_variants/$v/librtx_hip.so; fi
  echo "== ${v:-default}"
  timeout -k 10 100 python scripts/run_configs.py c2photon --repeat 2 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_render'], d['film_mean_clamped'])"
done
