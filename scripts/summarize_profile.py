"""Condense a scripts/profile.sh output directory into the files kept under profiles/.

  python scripts/summarize_profile.py gpurun_out/<dir> profiles/<dir> [kernel-name substring] [--traffic-json]
Writes kernel_stats.csv (copied from the --kernel-trace --stats run) and pmc_per_launch.json (per-launch
means of every counter for the kernels whose name contains the substring).  --traffic-json also rewrites
profiles/pmc_traffic.json, which bench.py reads for roofline.traffic."""
import csv, glob, json, os, shutil, sys
src, dst = sys.argv[1], sys.argv[2]
pat = sys.argv[3] if len(sys.argv) > 3 and not sys.argv[3].startswith("--") else "render_fused_kernel"
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv")) + glob.glob(os.path.join(src, "*", "*kernel_stats.csv")):
    shutil.copy(f, os.path.join(dst, "kernel_stats.csv")); break
sums, launches, kinfo = {}, {}, {}
for f in glob.glob(os.path.join(src, "pmc_*", "*", "*counter_collection.csv")):
    ids = set()
    for r in csv.DictReader(open(f)):
        if pat not in r["Kernel_Name"]: continue
        sums[r["Counter_Name"]] = sums.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        ids.add(r["Dispatch_Id"])
        kinfo = {k: r[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size") if k in r}
        name = r["Counter_Name"]
        launches.setdefault(name, set()).add(r["Dispatch_Id"])
out = {"_kernel": kinfo, "_kernel_name_contains": pat}
for k, v in sorted(sums.items()): out[k] = v / max(1, len(launches[k]))
json.dump(out, open(os.path.join(dst, "pmc_per_launch.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
if "--traffic-json" in sys.argv and "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    old = json.load(open(os.path.join(root, "profiles", "pmc_traffic.json")))
    old.update({"hbm_bytes_per_launch": (2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024.0, "fetch_size_kib": out["FETCH_SIZE"], "write_size_kib": out["WRITE_SIZE"],
                "note": "render_fused_kernel<float,FLAT>; separate --pmc passes (%s); FETCH_SIZE doubled per the gfx950 correction" % dst})
    json.dump(old, open(os.path.join(root, "profiles", "pmc_traffic.json"), "w"), indent=1)
