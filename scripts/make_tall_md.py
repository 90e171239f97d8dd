#!/usr/bin/env python3
"""Turns gpurun_out/r3w/tall.md (scripts/gpu_r3w.sh) into the first part of profiles/r03_tall_columns.md."""
import re, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
txt = open(os.path.join(ROOT, "gpurun_out", "r3w", "tall.md")).read()
secs = re.split(r'^## ', txt, flags=re.M)[1:]
def parse(sec):
    d = {}
    for ln in sec.splitlines():
        m = re.match(r'\| ([^|]+) \| (\d+) \| ([\d.]+) \| (\d+) \| (\d+) \| ([\d.]+) \|', ln)
        if m: d[(m.group(1).strip(), int(m.group(2)))] = (float(m.group(3)), float(m.group(6)))
    return d
new, old, f64 = parse(secs[0]), parse(secs[1]), parse(secs[2])
names = ["SimplexBijector", "inverse(SimplexBijector)", "OrderedBijector", "vjp(SimplexBijector)", "vjp(inverse(SimplexBijector))"]
out = []
out.append("| K | " + " | ".join(n + " G-lane / walkers" for n in names) + " |")
out.append("|---|" + "---|" * len(names))
for K in (100, 160, 200, 256, 300, 500, 1000, 2000):
    out.append(f"| {K} | " + " | ".join("**%.1f** / %.1f" % (new[(n, K)][1], old[(n, K)][1]) for n in names) + " |")
out.append("")
out.append("| K (Float64, 2^19 columns) | " + " | ".join(names) + " |")
out.append("|---|" + "---|" * len(names))
for K in (100, 200, 500, 1000):
    out.append(f"| {K} | " + " | ".join("%.1f" % f64[(n, K)][1] for n in names) + " |")
print("\n".join(out))
