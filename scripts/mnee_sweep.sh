#!/bin/bash
# Launch-geometry sweep of the batched manifold-NEE kernel (work items per lane; 0 = the resident grid), then occupancy variants.
set -o pipefail
OUT=gpurun_out/mnee_prof; mkdir -p $OUT; : > $OUT/sweep.jsonl
for n in 0 4 8 16 32 64; do
  RTX_ITEMS_PER_LANE=$n timeout -k 10 120 python scripts/ab_render.py c2mnee -- - 2>/dev/null | grep '"lib"' | sed "s/^{/{\"items_per_lane\": $n, /" >> $OUT/sweep.jsonl || exit 1
done
timeout -k 10 200 python scripts/ab_render.py c2mnee -- lib_variants/mnee_w3.so - lib_variants/mnee_w5.so 2>/dev/null | grep '"lib"' >> $OUT/sweep.jsonl
cat $OUT/sweep.jsonl | cut -c1-200
