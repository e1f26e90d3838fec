#!/bin/bash
# Memory-pipeline counters of the traversal kernels on the 10 M-triangle configuration (one rocprofv3 --pmc run per group).
#   scripts/pmc_c5.sh [out dir under gpurun_out]   (RTX_HIP_LIB selects a library variant)
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/${1:-pmc_c5}; mkdir -p $OUT
ARGS="--config c5 --steps 1 --warmup 0 --cpu-baseline-seconds 0 --no-extras --spp 16"
for pass in "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE" \
            "TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TA_TOTAL_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
            "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
            "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 200 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$tag -- python3 bench.py $ARGS > $OUT/pmc_$tag.log 2>&1 || { echo "pmc pass $tag failed"; tail -5 $OUT/pmc_$tag.log; exit 1; }
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections, re, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(sys.argv[1] + "/pmc_*/*/*counter_collection.csv"):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void rtx::", "")
        if "wf_" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"], k) not in seen: seen.add((r["Dispatch_Id"], k)); calls[k] += 1
out = {k: dict(v, dispatches=calls[k]) for k, v in agg.items()}
json.dump(out, open(sys.argv[1] + "/pmc_summary.json", "w"), indent=1)
for k, v in out.items():
    print(k, {a: f"{b:.3e}" for a, b in sorted(v.items())})
PY
