#!/usr/bin/env python3
"""Where the vector instructions of the headline kernel go, per phase of a path vertex.

Static side: the flat lean instance of render_fused_kernel is compiled for gfx950 with line tables, disassembled, and every
instruction is attributed through its full inline stack (llvm-symbolizer --inlines) to a phase of the loop body.
Dynamic side: a branch of a wave costs its instructions whenever ONE lane takes it, so each phase is weighted with the share of
wave-trips in which some lane took it, measured on the GPU by the diagnostic build (make -C ray-tracer_amd/csrc phase_stats;
python scripts/ab_render.py c2 64 -- ray-tracer_amd/lib/librtx_hip_phase_stats.so 2> log), and loop bodies with their trip counts
(the scene's spheres and planes).

Usage: python scripts/phase_table.py [phase_stats.log] [--spheres 3 --planes 5 --lights 1]     (needs hipcc; no GPU)"""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "ray-tracer_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
KERNEL = "_ZN3rtx19render_fused_kernelIfLi2ELb1ELi0ELi31EEE"
FLAGS = ("--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops -fno-hip-fp32-correctly-rounded-divide-sqrt "
         "-funsafe-math-optimizations -fgpu-flush-denormals-to-zero -gline-tables-only -c --cuda-device-only").split()

args = sys.argv[1:]
opt = {"--spheres": 3, "--planes": 5, "--lights": 1}
for k in list(opt):
    if k in args:
        i = args.index(k)
        opt[k] = int(args[i + 1])
        del args[i:i + 2]
stats_log = args[0] if args else None

tmp = tempfile.mkdtemp(prefix="phase_table_")
obj, co = os.path.join(tmp, "k32.o"), os.path.join(tmp, "k32.co")
subprocess.run(["hipcc", *FLAGS, os.path.join(CS, "kernels_f32.hip"), "-o", obj], check=True, stderr=subprocess.DEVNULL)
subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={obj}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
syms = subprocess.run([f"{LLVM}/llvm-objdump", "-t", co], capture_output=True, text=True, check=True).stdout
name = next(l.split()[-1] for l in syms.split("\n") if KERNEL in l and " F " in l)
dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co, f"--disassemble-symbols={name}"], capture_output=True, text=True, check=True).stdout
ins = [(int(m.group(2), 16), m.group(1)) for m in (re.match(r"\s+(\S+)\s.*//\s*([0-9A-F]+):", l) for l in dis.split("\n")) if m]
sym = subprocess.run([f"{LLVM}/llvm-symbolizer", f"--obj={co}", "--inlines", "--functions=short", "--output-style=LLVM"], input="\n".join(hex(a) for a, _ in ins),
                     capture_output=True, text=True, check=True).stdout.strip().split("\n\n")
assert len(sym) == len(ins)

src = {f: open(os.path.join(CS, f)).read().split("\n") for f in ("pt_device.hpp", "pt_kernels.hpp")}


def text(loc):
    f, ln = loc.split(":")[0], int(loc.split(":")[1])
    return src[f][ln - 1] if f in src and 0 < ln <= len(src[f]) else ""


def first_line(f, needle, after=1):
    for i in range(after - 1, len(src[f])):
        if needle in src[f][i]:
            return i + 1
    raise SystemExit(f"anchor not found: {needle}")


# anchors inside path_vertex_impl (pt_device.hpp), in source order
pv = first_line("pt_device.hpp", "RTX_DEV uint32_t path_vertex_impl(")
anchors = [("settle", "if (pending) {"), ("depth_miss", "if (bounce >= P.max_depth)"), ("shaded", "did |= STEP_SHADED;"), ("area", "if (rec.mat < 0) {  // area light"),
           ("material", "const DMaterial<R>& m = mats[rec.mat];"), ("rng", "const U4 rn = rng_block("), ("nee", "if (m.type == MAT_LAMBERTIAN) {"),
           ("scatter", "sdir = rec.n + unit_vector("), ("metal", "} else if (m.type == MAT_METAL) {"), ("glass", "} else {  // dielectric"), ("state", "ps.thr = ps.thr * atten;")]
bounds, at = [], pv
for key, needle in anchors:
    at = first_line("pt_device.hpp", needle, at)
    bounds.append((at, key))
loop_first = first_line("pt_kernels.hpp", "while (active) {")
loop_last = first_line("pt_kernels.hpp", "w_samples += (uint32_t)__popcll(__ballot(started));", loop_first)
end_first = first_line("pt_kernels.hpp", "bool started = false;", loop_first)

PHASES = {  # key: (label, weight key, loop key)
    "closest_pre": ("closest hit: set-up, winner's record, point and normal", "trip", None),
    "closest_sphere": ("closest hit: one sphere", "trip", "--spheres"),
    "closest_plane": ("closest hit: one plane", "trip", "--planes"),
    "settle": ("a BSDF ray reached a light: its pdf as a light sample, MIS weight", "settle", None),
    "settle_light": ("... light_pdf, per light", "settle", "--lights"),
    "depth_miss": ("depth limit, miss, background", "miss", None),
    "shaded": ("vertex bookkeeping", "trip", None),
    "area": ("area-light arrival (no such light in this scene)", "area", None),
    "material": ("material record, albedo, emission", "trip", None),
    "rng": ("the vertex's random block (pcg4d)", "trip", None),
    "nee": ("Lambertian: light sample, pdfs, MIS weight, contribution", "lambert", None),
    "nee_light": ("... sample_light, per light", "lambert", "--lights"),
    "shadow_pre": ("shadow query: set-up", "shadow", None),
    "shadow_sphere": ("shadow query: one sphere", "shadow", "--spheres"),
    "shadow_plane": ("shadow query: one plane", "shadow", "--planes"),
    "scatter": ("Lambertian: cosine scatter", "lambert", None),
    "metal": ("metal: reflect (+ fuzz)", "metal", None),
    "glass": ("glass: Schlick, reflect / refract", "glass", None),
    "state": ("throughput, Russian roulette, next ray and flags", "trip", None),
    "end": ("path ended: chunk sum, next work item, camera ray with its random block", "ended", None),
    "stats": ("statistics (ballots), loop control", "trip", None),
}


def classify(frames):
    k = next((int(loc.split(":")[1]) for f, loc in frames if f.startswith("render_fused_kernel")), None)
    if k is None or not (loop_first <= k <= loop_last):
        return None
    names = [f for f, _ in frames]
    for f, loc in frames:
        if f.startswith("flat_closest_tk"):
            t = text(loc)
            return "closest_sphere" if "flat_sphere_t" in t or "n_prims" in t else "closest_plane" if "flat_plane_pd" in t or "n_planes" in t else "closest_pre"
        if f.startswith("flat_occluded"):
            t = text(loc)
            return "shadow_sphere" if "flat_sphere_t" in t else "shadow_plane" if "flat_plane_pd" in t else "shadow_pre"
    if any(n.startswith(("flat_closest", "flat_hit_from", "trace_closest")) for n in names):
        return "closest_pre"
    if any(n.startswith("trace_occluded") for n in names) or any("lambda" in n and "operator()" in n for n in names):
        return "shadow_pre"
    if any(n.startswith("path_begin") for n in names) or k >= end_first:
        return "end"
    v = next((int(loc.split(":")[1]) for f, loc in frames if f.startswith("path_vertex_impl")), None)
    if v is None:
        return "stats"
    key = "settle"
    for line, name_ in bounds:
        if v >= line:
            key = name_
    if key == "settle" and any(n.startswith("light_pdf") for n in names):
        return "settle_light"
    if key == "nee" and any(n.startswith("sample_light") for n in names):
        return "nee_light"
    return key


valu, salu, outside = Counter(), Counter(), 0
for (addr, op), block in zip(ins, sym):
    ls = block.strip().split("\n")
    frames = [(ls[i], ls[i + 1].split("/")[-1]) for i in range(0, len(ls) - 1, 2)]
    ph = classify(frames)
    if ph is None:
        outside += op.startswith("v_")
        continue
    if op.startswith("v_"):
        valu[ph] += 1
    elif op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop")):
        salu[ph] += 1

weights = {"trip": 1.0}
lanes = {}
if stats_log:
    line = next(l for l in open(stats_log) if l.startswith("phase stats:"))
    st = {int(m.group(1)): (int(m.group(2)), int(m.group(3))) for m in re.finditer(r"(\d+):(\d+)/(\d+)", line)}
    trips = st[12][0]
    slot = {"shadow": 0, "settle": 3, "miss": 4, "area": 5, "lambert": 6, "metal": 7, "glass": 8, "roulette": 9, "ended": 13}
    for k, b in slot.items():
        weights[k] = st[b][0] / trips
        lanes[k] = st[b][1] / (64.0 * st[b][0]) if st[b][0] else 0.0
    lanes["trip"] = st[12][1] / (64.0 * trips)
    weights["miss"] = (st[4][0] + st.get(11, (0, 0))[0]) / trips
else:
    weights.update({k: 1.0 for k in ("shadow", "settle", "miss", "area", "lambert", "metal", "glass", "roulette", "ended")})

print("| phase | VALU (static) | x loop trips | x wave-trips that run it | VALU per wave-trip | lanes busy in it | SALU (static) |")
print("|---|---|---|---|---|---|---|")
tot = tot_salu = 0.0
for key, (label, wk, loop) in PHASES.items():
    n = opt[loop] if loop else 1
    dyn = valu[key] * n * weights[wk]
    tot += dyn
    tot_salu += salu[key] * n * weights[wk]
    busy = f"{lanes[wk]:.2f}" if wk in lanes else "-"
    print(f"| {label} | {valu[key]} | {n} | {weights[wk]:.3f} | {dyn:.0f} | {busy} | {salu[key]} |")
print(f"| **sum** | {sum(valu.values())} | | | **{tot:.0f}** | | {sum(salu.values())} (weighted {tot_salu:.0f}) |")
print(f"\n(instructions of the kernel outside the loop: {outside} VALU)")
