#!/bin/bash
# Runs scripts/bin/gather_calib once per pattern under the memory-side counters and prints counter bytes against known bytes.
set -o pipefail
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
for spec in stream:4096 gather64:4096 gather128:4096 gather64x4:4096 gather64:800 gather64x4:800 gather64x4:128; do
  pat=${spec%%:*}; mib=${spec##*:}
  D=$(mktemp -d /tmp/calib.XXXXXX)
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --output-format csv -d $D -- scripts/bin/gather_calib $pat $mib > $D/log.txt 2>&1 || { echo "pmc run failed ($pat)"; tail -5 $D/log.txt; exit 1; }
  grep '^{' $D/log.txt
  python3 - $D $pat <<'PY'
import csv, glob, sys, collections, json
agg = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "gather_kernel" in r["Kernel_Name"] or "stream_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
known = None
for l in open(sys.argv[1] + "/log.txt"):
    if l.startswith("{"): known = json.loads(l)["known_bytes"]
fs = agg.get("FETCH_SIZE", 0.0) * 1024.0
rq = agg.get("TCC_EA0_RDREQ_sum", 0.0)
print(json.dumps({"pattern": sys.argv[2], "known_bytes": known, "FETCH_SIZE_bytes": fs, "FETCH_SIZE_over_known": fs / known, "TCC_EA0_RDREQ": rq,
                  "known_bytes_per_request": known / rq if rq else None}))
PY
  rm -rf $D
done
