#!/usr/bin/env python3
"""Static instruction breakdown of one kernel from hipcc's assembly (-S -gline-tables-only): VALU / SALU / memory instructions
per source function, attributed by the .loc line of the innermost inlined frame.  Usage:
    python scripts/isa_breakdown.py <file.s> <kernel-symbol-prefix> [--loop]    (--loop: only blocks inside loops)"""
import re
import sys
from collections import Counter

path, prefix = sys.argv[1], sys.argv[2]
only_loop = "--loop" in sys.argv
# function line ranges of pt_device.hpp / pt_kernels.hpp, from the sources themselves
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "ray-tracer_amd", "csrc")


def function_ranges(fname):
    out, cur, start = [], None, 0
    pat = re.compile(r"^(?:template <[^>]*>\s*)?(?:RTX_DEV|__global__|static|inline)?.*?\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;]*$")
    lines = open(os.path.join(CS, fname)).read().split("\n")
    for i, l in enumerate(lines, 1):
        if l and not l.startswith((" ", "\t", "}", "#", "/")) and "(" in l and not l.rstrip().endswith(";"):
            m = pat.match(l)
            if m:
                if cur:
                    out.append((start, i - 1, cur))
                cur, start = m.group(1), i
    if cur:
        out.append((start, len(lines), cur))
    return out


ranges = {f: function_ranges(f) for f in ("pt_device.hpp", "pt_kernels.hpp", "pt_mnee.hpp")}
files = {}
counts, salu, mem = Counter(), Counter(), Counter()
inside, depth_ok, cur = False, True, ("?", 0)
for line in open(path):
    m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', line)
    if m:
        files[int(m.group(1))] = os.path.basename(m.group(2))
        continue
    if line.startswith(prefix) and line.rstrip().endswith(("@" + line.split(":")[0]).strip()) or (line.startswith(prefix) and ":" in line and not inside):
        inside = True
        continue
    if not inside:
        continue
    if line.startswith(".Lfunc_end"):
        break
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
    if m:
        cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
        continue
    if line.startswith(".LBB"):
        depth_ok = ("Loop" in line or "Depth" in line) if only_loop else True
        continue
    t = line.strip()
    if not t or t.startswith((";", ".")) or not depth_ok:
        continue
    op = t.split()[0]
    f, ln = cur
    name = f
    for a, b, fn in ranges.get(f, []):
        if a <= ln <= b:
            name = fn
            break
    if op.startswith("v_"):
        counts[name] += 1
    elif op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop")):
        salu[name] += 1
    elif op.startswith(("global_", "ds_", "buffer_", "flat_", "scratch_")):
        mem[name] += 1
tot = sum(counts.values())
print(f"{'function':28s} {'VALU':>6s} {'%':>6s} {'SALU':>6s} {'mem':>5s}")
for name, c in counts.most_common():
    print(f"{name:28s} {c:6d} {100.0 * c / tot:6.1f} {salu[name]:6d} {mem[name]:5d}")
print(f"{'total':28s} {tot:6d} {'':6s} {sum(salu.values()):6d} {sum(mem.values()):5d}")
