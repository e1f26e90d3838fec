#!/bin/bash
# scripts/pmc_any.sh "<counters>" <python script and args...>  -> per-kernel sums of the counters on stdout
set -o pipefail
export TMPDIR=/tmp
C="$1"; shift
D=$(mktemp -d /tmp/pmc.XXXXXX)
timeout -s KILL 300 rocprofv3 --pmc $C --output-format csv -d $D -- python3 "$@" > $D/log.txt 2>&1 || { echo "pmc run failed"; tail -5 $D/log.txt; exit 1; }
python3 - $D <<'PY'
import csv, glob, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0][-30:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"], k) not in seen: seen.add((r["Dispatch_Id"], k)); calls[k] += 1
for k, v in agg.items():
    if "rocclr" not in k: print(k, "calls", calls[k], {a: int(b) for a, b in sorted(v.items())})
PY
rm -rf $D
