"""Turns the scripts/profile.sh summaries under gpurun_out/prof_<config>/ into profiles/bench_profiles.json, the file bench.py
takes its counter-derived roofline figures from, and copies each summary (and its rocprofv3 kernel statistics) to
profiles/r03_prof_<config>.json (+ _kernel_stats.csv).

  python scripts/make_bench_profiles.py [c2 c3 c5]
Every entry names the workload and the SHA-256 of the library that was profiled; bench.py uses an entry only when both match
what it has just timed, so this runs after the last rebuild of ray-tracer_amd/lib/librtx_hip.so."""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import CONFIGS, sha256_of

dst = os.path.join(ROOT, "profiles", "bench_profiles.json")
allp = json.load(open(dst)) if os.path.exists(dst) else {}
lib = sha256_of(os.path.join(ROOT, "ray-tracer_amd", "lib", "librtx_hip.so"))
for cfg in (sys.argv[1:] or ["c2", "c3", "c5"]):
    src = os.path.join(ROOT, "gpurun_out", f"prof_{cfg}", "summary.json")
    if not os.path.exists(src):
        print(f"{cfg}: no {src}")
        continue
    s = json.load(open(src))
    if s.get("lib_sha256") != lib:
        print(f"{cfg}: the summary names library {str(s.get('lib_sha256'))[:12]}, the tree holds {lib[:12]}: skipped")
        continue
    renders = s["renders_profiled"]
    name, W, H, spp, depth = CONFIGS[cfg][0]
    want = "wf_extend" if cfg == "c5" else "render_fused_kernel"
    cands = [(k, e) for k, e in s["kernels"].items() if k.startswith(want) and "counters_sum" in e]
    if not cands:
        print(f"{cfg}: no {want} kernel in the summary")
        continue
    k, e = max(cands, key=lambda kv: kv[1].get("total_ms", 0.0))
    c, d = e["counters_sum"], e.get("derived", {})
    allp[cfg] = {
        "workload": [name, W, H, spp, depth, 1], "lib_sha256": lib, "kernel": k, "renders_profiled": renders,
        "launches_per_render": e["dispatches"] / renders,
        # sums over the kernel's launches of ONE render of the workload
        "SQ_INSTS_VALU": c["SQ_INSTS_VALU"] / renders, "hbm_bytes_per_launch": d.get("hbm_bytes_scattered_64B" if cfg == "c5" else "hbm_bytes", 0.0) / renders or None,
        "hbm_bytes_rule": "FETCH_SIZE as tallied (64-byte requests: scattered 64-byte records) + WRITE_SIZE" if cfg == "c5" else "2 x FETCH_SIZE (128-byte requests tallied at 64) + WRITE_SIZE",
        "lanes_active": d.get("lanes_active"), "valu_busy": d.get("valu_busy"), "valu_dual_issue_share": d.get("valu_dual_issue_share"), "l2_hit_rate": d.get("l2_hit_rate"), "kernel_ms_rocprof": e.get("ms_per_render"),
        "source": f"profiles/r03_prof_{cfg}.json (rocprofv3 --kernel-trace --stats and separate --pmc passes of: bench.py {s.get('bench_args')})",
    }
    shutil.copy(src, os.path.join(ROOT, "profiles", f"r03_prof_{cfg}.json"))
    import glob
    for ks in glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{cfg}", "trace", "*", "*_kernel_stats.csv"))[:1]:
        shutil.copy(ks, os.path.join(ROOT, "profiles", f"r03_prof_{cfg}_kernel_stats.csv"))
    print(cfg, json.dumps(allp[cfg])[:400])
json.dump(allp, open(dst, "w"), indent=1)
