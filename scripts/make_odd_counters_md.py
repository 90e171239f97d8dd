"""profiles/r03_odd_counters.md from the two PMC passes of scripts/probe_odd_traffic.py (scripts/gpu_r3bj.sh): HBM bytes of the group
kernels at a whole-pack height and the odd height next to it.  usage: make_odd_counters_md.py <FETCH csv> <WRITE csv> <dims,comma>"""
import csv, re, sys

MAIN = ("planar_reg", "planar_walk", "planar_kernel", "planar_tile", "radial_kernel", "radial_walk", "colgroup", "colwalk", "chain_col",
        "chain_flat", "chain_tiny", "stacked_mixed", "stacked_tiny")


def load(p):
    out = []
    for r in csv.DictReader(open(p)):
        n = re.sub(r"\(anonymous namespace\)::|bjx::|void ", "", r["Kernel_Name"])
        n = re.sub(r"\(.*", "", n)
        if any(k in n for k in MAIN):
            out.append((int(r["Dispatch_Id"]), n, float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0))
    return sorted(out)


F, W = load(sys.argv[1]), load(sys.argv[2])
dims = [int(v) for v in sys.argv[3].split(",")]
N = 1 << 20
per = len(F) // len(dims)
print("| rows | kernel | µs (under the counter pass) | FETCH_SIZE ×2, KiB | WRITE_SIZE, KiB | read / input | written / output |")
print("|---|---|---|---|---|---|---|")
for i in range(0, len(F), 2):                       # two calls per case: the second one
    (_, n, f, t), (_, n2, w, _) = F[i + 1], W[i + 1]
    assert n == n2
    d = dims[i // per]
    alg_in, alg_out = d * 4 * N / 1024.0, (d * 4 + 4) * N / 1024.0
    print(f"| {d} | `{n}` | {t:.0f} | {2 * f:.0f} | {w:.0f} | {2 * f / alg_in:.3f} | {w / alg_out:.3f} |")
