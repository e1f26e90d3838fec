"""Condense a scripts/profile.sh output directory.

  python scripts/prof_summary.py gpurun_out/<dir> <summary.json>
Per kernel (short name): dispatches and the SUM of every counter over them, from the --pmc passes, plus calls / total / average
duration from the --kernel-trace --stats pass.  bench.py ran `--steps 2 --warmup 1`, i.e. three renders of the workload, so
per-render figures are sums / 3."""
import collections, csv, glob, json, os, re, sys
src, dst = sys.argv[1], sys.argv[2]
RENDERS = 3

def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?(?:rtx::)?([A-Za-z_0-9:]+)(<[^(]*>)?\(", name)
    if not m: return name[:60]
    return m.group(1) + (m.group(2) or "")

out = {"renders_profiled": RENDERS, "bench_args": open(os.path.join(src, "bench_args.txt")).read().strip() if os.path.exists(os.path.join(src, "bench_args.txt")) else None,
       "lib_sha256": open(os.path.join(src, "lib_sha256.txt")).read().strip() if os.path.exists(os.path.join(src, "lib_sha256.txt")) else None, "kernels": {}}
K = out["kernels"]
for f in glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if "rocclr" in r["Name"] or "at::native" in r["Name"]: continue
        K.setdefault(short(r["Name"]), {}).update({"calls": int(r["Calls"]), "total_ms": float(r["TotalDurationNs"]) / 1e6, "avg_ms": float(r["AverageNs"]) / 1e6,
                                                   "ms_per_render": float(r["TotalDurationNs"]) / 1e6 / RENDERS, "percent": float(r["Percentage"])})
for f in glob.glob(os.path.join(src, "pmc_*", "*", "*counter_collection.csv")):
    sums, ids, info = collections.defaultdict(lambda: collections.defaultdict(float)), collections.defaultdict(set), {}
    for r in csv.DictReader(open(f)):
        if "rocclr" in r["Kernel_Name"] or "at::native" in r["Kernel_Name"]: continue
        k = short(r["Kernel_Name"])
        sums[k][r["Counter_Name"]] += float(r["Counter_Value"])
        ids[k].add(r["Dispatch_Id"])
        info[k] = {x: r[x] for x in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size") if x in r}
    for k, v in sums.items():
        e = K.setdefault(k, {})
        e.setdefault("counters_sum", {}).update(v)
        e["dispatches"] = len(ids[k])
        e["resources"] = info[k]
# derived figures
for k, e in K.items():
    c = e.get("counters_sum", {})
    d = {}
    if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"] > 0:
        d["valu_issue_busy_4cyc"] = c["SQ_INSTS_VALU"] * 4.0 / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0)
    if "SQ_ACTIVE_INST_VALU" in c and c.get("GRBM_GUI_ACTIVE", 0) > 0:
        d["valu_busy"] = c["SQ_ACTIVE_INST_VALU"] / 256.0 / (c["GRBM_GUI_ACTIVE"] / 8.0)   # rocprofv3's VALUBusy: above 1 when instructions issue in pairs
    if "SQ_ACTIVE_INST_VALU2" in c and c.get("SQ_ACTIVE_INST_VALU", 0) > 0:
        d["valu_dual_issue_share"] = c["SQ_ACTIVE_INST_VALU2"] / c["SQ_ACTIVE_INST_VALU"]
    if "SQ_THREAD_CYCLES_VALU" in c and c.get("SQ_ACTIVE_INST_VALU", 0) > 0:
        d["lanes_active"] = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0)
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # FETCH_SIZE = 64 bytes x read requests at the L2's memory side, whatever their size.  A streaming read makes 128-byte
        # requests (double it); the tree walk's scattered 64-byte records make 64-byte requests (take it as it is): both measured
        # with scripts/gather_calib.sh on tables of known size.
        d["hbm_bytes"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        d["hbm_bytes_scattered_64B"] = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        if e.get("total_ms"):
            d["hbm_GBps"] = d["hbm_bytes"] / (e["total_ms"] * 1e-3) / 1e9
            d["hbm_GBps_scattered_64B"] = d["hbm_bytes_scattered_64B"] / (e["total_ms"] * 1e-3) / 1e9
    if "TCC_HIT_sum" in c and c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0) > 0:
        d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if "SQ_WAIT_ANY" in c and c.get("SQ_WAVE_CYCLES", 0) > 0:
        d["wave_cycles_waiting"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
        d["wave_cycles_issue_stalled"] = c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]
    if d: e["derived"] = d
json.dump(out, open(dst, "w"), indent=1)
for k, e in sorted(K.items(), key=lambda kv: -kv[1].get("total_ms", 0)):
    print(k[:70], {x: (round(v, 4) if isinstance(v, float) else v) for x, v in e.items() if x in ("calls", "ms_per_render", "percent")}, {x: round(v, 4) for x, v in e.get("derived", {}).items()})
