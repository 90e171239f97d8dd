"""profiles/r03_tall_columns.md, second part: the flows / elementwise rows of profiles/r02_tall_columns.md (dim = 101 ... 1000, 2^22 columns)
next to this round's numbers.  usage: make_tall_flows_md.py <new table from scripts/bench_small_dims.py> [<more tables> ...]"""
import re
import sys


def rows(path, skip_simplex=True):
    out = {}
    for ln in open(path):
        m = re.match(r"\|\s*(.+?)\s*\|\s*(\d+)\s*\|\s*([\d.]+)\s*\|\s*(\d+)\s*\|\s*(\d+)\s*\|\s*([\d.]+)\s*\|", ln)
        if not m:
            continue
        name, dim, ms, bps, gbs, pct = m.group(1), int(m.group(2)), float(m.group(3)), int(m.group(4)), int(m.group(5)), float(m.group(6))
        if skip_simplex and ("Simplex" in name or "Ordered" in name):
            continue
        out[(name, dim)] = (ms, bps, gbs, pct)
    return out


old = rows("profiles/r02_tall_columns.md")
new = {}
for p in sys.argv[1:]:
    new.update(rows(p))
dims = sorted({d for (_, d) in new})
names = []
for (n, _) in new:
    if n not in names:
        names.append(n)
print("| bijector | dim | kernel ms (2^22 columns) | alg. B/sample | GB/s | % of 8 TB/s | round 2 |")
print("|---|---|---|---|---|---|---|")
below = 0
for d in dims:
    for n in names:
        if (n, d) not in new:
            continue
        ms, bps, gbs, pct = new[(n, d)]
        o = old.get((n, d))
        below += pct < 50.0
        print(f"| {n} | {d} | {ms:.4f} | {bps} | {gbs} | {'**%.1f**' % pct if pct >= 50.0 else '%.1f' % pct} | {('%.1f' % o[3]) if o else ''} |")
print()
print(f"{len(new) - below} of {len(new)} rows at or above 50 % of the 8 TB/s peak.")
