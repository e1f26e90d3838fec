#!/bin/bash
# PMC passes for the 10 M triangle configuration at reduced spp.  Usage: scripts/pmc_c5.sh <dir under gpurun_out> [run_configs args]
set -o pipefail
OUT=gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 300 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$tag -- python3 scripts/run_configs.py c5 --repeat 1 "$@" > $OUT/pmc_$tag.log 2>&1 || { echo "pmc pass $tag failed"; tail -5 $OUT/pmc_$tag.log; exit 1; }
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        import re
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0][-36:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    if 'rocclr' in k: continue
    print(k, {a: (int(b) if b > 1e6 else b) for a, b in sorted(v.items())})
PY
echo pmc_done
