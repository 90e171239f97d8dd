"""profiles/<TAG>_bench_repeats.md from the profile set's line and gpurun_out/bench_rep_*.json (one `python bench.py > gpurun_out/bench_rep_N.json`
per gpurun call = one fresh box each):   python scripts/make_bench_repeats.py r06"""
import glob
import json
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BARS = {"c2": 9370, "c3": 9300, "c3_uncached": 9300, "c4": 4670, "c5a": 9370, "c5b": 196}


def line(path):
    txt = [l for l in open(path).read().splitlines() if l.startswith("{")]
    return json.loads(txt[-1]) if txt else None


def cells(d):
    rows = {r["workload"]: r for r in d.get("rows", [])}
    sc = {s.get("name", ""): s for s in d.get("small_calls", [])} if isinstance(d.get("small_calls"), list) else {}
    g = lambda w, k="value": rows.get(w, {}).get(k)
    small = [s.get("us_per_call") for s in d.get("small_calls", [])] if isinstance(d.get("small_calls"), list) else []
    return {"c2": d["value"], "frac": d["roofline"]["frac"], "c3": g("c3"), "c3_frac": g("c3", "frac"), "c3_uncached": g("c3_uncached"), "c4": g("c4"), "c5a": g("c5a"),
            "c5b": g("c5b"), "c2_f64": g("c2_f64"), "c4_f64": g("c4_f64"), "c1_us": (g("c1", "ms_per_step") or 0) * 1e3, "small": small}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    boxes = [(f"profile {tag} (first thing on its box)", line(os.path.join(R, "profiles", f"{tag}_bench_default.json")))]
    for p in sorted(glob.glob(os.path.join(R, "gpurun_out", "bench_rep_*.json"))):
        d = line(p)
        if d:
            boxes.append((f"fresh box {os.path.basename(p)[10:-5]}", d))
    n = len(boxes)
    out = [f"# {tag} — the driver's line (`python bench.py`, no flags) on {n} fresh boxes with the shipped library (`profiles/{tag}_lib_sha16.txt`)", "",
           "One gpurun call per row = one fresh box; nothing ran on the box before the line.  M samples/s unless noted; bars of BASELINE.md: C2 ≥ 9 370, C3 ≥ 9 300, C4 ≥ 4 670, C5a ≥ 9 370, C5b ≥ 196.",
           "", "| box | C2 headline | frac | C3 (table kept) | C3 frac | C3 (default: table rebuilt) | C4 | C5a | C5b | C2 f64 | C4 f64 | C1 µs/call | small calls µs |", "|---|" + "---|" * 12]
    cs = []
    for name, d in boxes:
        c = cells(d)
        cs.append(c)
        f = lambda v, nd=0: "—" if v is None else (f"{v:.{nd}f}")
        out.append(f"| {name} | {f(c['c2'])} | {f(c['frac'], 3)} | {f(c['c3'])} | {f(c['c3_frac'], 3)} | {f(c['c3_uncached'])} | {f(c['c4'])} | {f(c['c5a'])} | {f(c['c5b'], 1)} | {f(c['c2_f64'])} | {f(c['c4_f64'])} | "
                   f"{f(c['c1_us'], 2)} | {' / '.join(f(s, 2) for s in c['small'])} |")
    under = {k: sum(1 for c in cs if c.get(k) is not None and c[k] < BARS[k]) for k in BARS}
    rng = lambda k: f"{min(c[k] for c in cs if c.get(k) is not None):.0f} … {max(c[k] for c in cs if c.get(k) is not None):.0f}"
    out += ["", f"Headline {rng('c2')}.  C3 with the table kept: {rng('c3')} (under its bar of 9 300 on {under['c3']} of {n} boxes); with the library's default policy (table rebuilt per call): "
            f"{rng('c3_uncached')} (under the bar on {under['c3_uncached']} of {n}).  Rows under their bar anywhere else: "
            + (", ".join(f"{k} on {v}" for k, v in under.items() if v and k not in ("c3", "c3_uncached")) or "none") + "."]
    path = os.path.join(R, "profiles", f"{tag}_bench_repeats.md")
    open(path, "w").write("\n".join(out) + "\n")
    print("\n".join(out[-3:]))


if __name__ == "__main__":
    main()
