#!/usr/bin/env python3
"""Turns gpurun_out/<TAG>/ (written by `scripts/gpu_run.sh profile <TAG>` on the GPU box) into the tracked
evidence under profiles/: <TAG>_bench_lines.jsonl, <TAG>_<wl>_kernel_stats.csv (top rows),
<TAG>_<wl>_pmc_{FETCH,WRITE}_SIZE.csv (dominant-kernel rows) and profiles/traffic.json.

HBM bytes per step = (2 * FETCH_SIZE + WRITE_SIZE) KiB * 1024 summed over the dominant kernel
launches of one step — the x2 is the gfx950 FETCH_SIZE half-count of wide coalesced reads
(MI355X_MICROARCH.md, HBM section); it is exact for the 16-byte streaming loads these kernels use."""
import csv
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOMINANT = {"c1": ["chain_flat"], "c2": ["chain_flat_kernel"], "c2v": ["chain_flat_kernel"], "c2_f64": ["chain_flat_kernel"], "c4_f64": ["planar_mfma64_kernel"], "c3": ["rqs_lds_kernel"], "c4": ["planar_reg"],
            "c5a": ["quad_stream_kernel"], "c5b": ["chol_inv_chunk_kernel"], "vcorr": ["matrix_cyc_kernel<float, 8, 4, 0, false>"], "pdvec": ["matrix_cyc_kernel<float, 8, 4, 3, false>"]}   # the forward kernel is the timed one (the inverse builds the input)


def main(tag):
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    traffic_path = os.path.join(dst, "traffic.json")
    try:
        traffic = json.load(open(traffic_path))
    except Exception:
        traffic = {}
    traffic["_how"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), KiB per dispatch of the dominant "
                       "kernel(s); HBM bytes per step = (2*FETCH_SIZE + WRITE_SIZE)*1024 summed over the dominant launches of one step — the x2 is the "
                       "gfx950 FETCH_SIZE half-count correction of MI355X_MICROARCH.md (HBM section); raw rows in profiles/<tag>_<wl>_pmc_*.csv")
    lines = []
    for wl, subs in DOMINANT.items():
        bj = os.path.join(src, f"bench_{wl}.json")
        if os.path.exists(bj) and os.path.getsize(bj) > 2:
            lines.append(open(bj).read().strip())
        ks = os.path.join(src, f"{wl}_kernel_stats.csv")
        if os.path.exists(ks):
            rows = list(csv.reader(open(ks)))
            with open(os.path.join(dst, f"{tag}_{wl}_kernel_stats.csv"), "w", newline="") as f:
                csv.writer(f).writerows(rows[:12])
        per = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            p = os.path.join(src, f"{wl}_pmc_{c}.csv")
            if not os.path.exists(p):
                continue
            rows = [r for r in csv.DictReader(open(p)) if any(s in r["Kernel_Name"] for s in subs)]
            if not rows:
                continue
            with open(os.path.join(dst, f"{tag}_{wl}_pmc_{c}.csv"), "w", newline="") as f:
                w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
                w.writeheader()
                w.writerows(rows)
            by = defaultdict(list)
            for r in rows:
                by[r["Kernel_Name"].split("(")[0] + "|" + r["Kernel_Name"][:160]].append(float(r["Counter_Value"]))
            # one launch of every distinct dominant kernel per step (c3: forward + inverse)
            per[c] = {k: sum(v) / len(v) for k, v in by.items()}
        if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
            f_kib = sum(per["FETCH_SIZE"].values())
            w_kib = sum(per["WRITE_SIZE"].values())
            entry = {"tag": tag, "kernels": sorted(k.split("|")[1] for k in per["FETCH_SIZE"]), "FETCH_SIZE_KiB": f_kib, "WRITE_SIZE_KiB": w_kib,
                     "hbm_bytes_per_launch": (2 * f_kib + w_kib) * 1024.0}
            for ln in lines[-1:]:
                try:
                    d = json.loads(ln)
                    entry["algorithmic_bytes_per_launch"] = d["roofline"]["algorithmic_bytes_per_launch"]
                except Exception:
                    pass
            traffic[wl] = entry
    if lines:
        with open(os.path.join(dst, f"{tag}_bench_lines.jsonl"), "w") as f:
            f.write("\n".join(lines) + "\n")
    # which binary the evidence belongs to (tests/test_profiles_fresh.py: the kernel names of these files must exist in the .so of the tree)
    sha = os.path.join(src, "lib_sha16.txt")
    if os.path.exists(sha):
        traffic["_lib_sha16"] = open(sha).read().strip()
    traffic["_tag"] = tag
    json.dump(traffic, open(traffic_path, "w"), indent=1)
    for extra in ("bench_default.json", "bench_default_detail.json", "rows.md", "rows_kernel_stats.csv", "f64_rows.md", "f64math_bench.txt", "planar_mfma_ab.txt", "small_sizes.md",
                  "kernel_trace_timed.jsonl", "small_calls.md", "lds_atomics.txt", "c3_sq_pmc_1.txt", "c3_sq_pmc_2.txt", "c3_sq_pmc_3.txt", "c4_f64_sq_pmc_1.txt", "c4_f64_sq_pmc_2.txt", "c4_f64_sq_pmc_3.txt", "lib_sha16.txt", "lib_bytes.txt", "first_call.txt", "ordered_tall.md", "planar_heights.md", "small_dims.md", "host_overhead.txt", "c3_table_policy.txt"):
        pe = os.path.join(src, extra)
        if os.path.exists(pe) and os.path.getsize(pe) > 2:
            with open(pe) as fi, open(os.path.join(dst, f"{tag}_{extra}"), "w") as fo:
                fo.write(fi.read())
    pt = os.path.join(src, "pytest_gpu.txt")
    if os.path.exists(pt):
        ls = open(pt).readlines()
        keep = [l for l in ls if " passed" in l or " failed" in l or l.startswith("FAILED")] or ls[-5:]
        open(os.path.join(dst, f"{tag}_pytest_gpu_tail.txt"), "w").write("".join(keep))
    mv = os.path.join(src, "matrix_vjp_errors.jsonl")
    if os.path.exists(mv):
        write_matrix_vjp_table(mv, os.path.join(dst, f"{tag}_matrix_vjp_errors.md"), tag)
    ve = os.path.join(src, "vjp_errors.jsonl")
    if os.path.exists(ve):
        write_vjp_error_table(ve, os.path.join(dst, f"{tag}_vjp_errors.md"), tag)
    print(json.dumps({k: v for k, v in traffic.items() if k != "_how"}, indent=1)[:3000])


def write_vjp_error_table(jsonl, out_md, tag):
    """tests/_tol.py (flat_close) records the worst error of every pullback / density comparison of the GPU suites: one row per
    (family, dtype), the family being the record's label with its shape numbers taken out."""
    import re
    from collections import OrderedDict

    rows = [json.loads(l) for l in open(jsonl) if l.strip()]
    agg = OrderedDict()
    for r in rows:
        fam = re.sub(r"\b(dim|K|N|layers|dt|seed|uplo|lbar|inv|inverse)=[^ :,]+", "", r["what"])
        fam = re.sub(r"\s+", " ", fam).strip(" :,")
        a = agg.setdefault((fam, r["dtype"], r["per"], bool(r.get("conditioned"))), {"n": 0, "worst": 0.0, "what": "", "over": 0.0, "amp": None, "frac": 0.0, "flatcol": 0.0})
        a["n"] += 1
        if r["worst"] >= a["worst"]:
            a["worst"], a["what"] = r["worst"], r["what"]
            a["amp"] = r.get("amp_at_worst")
        a["over"] = max(a["over"], r.get("worst_over_allowed", r["worst_over_rtol"]))
        a["frac"] = max(a["frac"], r.get("frac_over_flat", 0.0))
        if r.get("worst_flat_column") is not None:
            a["flatcol"] = max(a["flatcol"], r["worst_flat_column"])
    out = [f"# {tag} — measured errors of every pullback / density comparison of the GPU suites (`tests/_tol.py`, GPU vs the FD-pinned oracle)", "",
           "Bar: north_star's FLAT relative tolerance, 1e-3 Float32 / 1e-6 Float64, on the stated scale — `sample`: max-norm of the reference cotangent of that column;",
           "`tensor`: max-norm of the (small) parameter-cotangent tensor; `element`: |ref| + 1 (log-densities).  No multiplier anywhere (`grep -c \"RTOL\\[dt\\] \\* [0-9]\" tests/test_gpu_*.py` = 0).",
           "Rows marked *conditioned* add, per column, `max(rtol, 4·a)` with a first-order amplification `a` computed by the test from the data and printed:",
           "Simplex in Float32: `a = sqrt(K)·eps/(2·min remainder)` (the running stick sum every Float32 evaluation carries, the reference's included);",
           "inverse PlanarLayer: `a = eps·(Π_l max(1, 1/d_l))²`, `d_l = 1 + wᵀû·sech²` the layer's determinant.  For those rows `worst / bar` is against the conditioned bar, the",
           "share of columns over the FLAT bar and the worst column that had no allowance are listed too.",
           f"Source: the pytest run of `profiles/{tag}_pytest_gpu_tail.txt` ({len(rows)} comparisons).", "",
           "| comparison | dtype | scale | checks | worst error | worst / bar | conditioned: amplification at worst, columns over flat, worst un-allowanced column |", "|---|---|---|---|---|---|---|"]
    for (fam, dt, per, cond), a in sorted(agg.items(), key=lambda kv: (kv[0][1], -kv[1]["over"])):
        extra = f"a = {a['amp']:.2e}, {100 * a['frac']:.1f} %, {a['flatcol']:.2e}" if cond and a["amp"] is not None else ""
        out.append(f"| {fam} | {dt} | {per} | {a['n']} | {a['worst']:.2e} | {a['over']:.3f} | {extra} |")
    for dt, bar in (("float32", 1e-3), ("float64", 1e-6)):
        flat = [a["worst"] for (f, d, p, c), a in agg.items() if d == dt and not c]
        if flat:
            out += ["", f"Worst {dt} on the flat bar: {max(flat):.2e} (bar {bar:.0e}); {sum(a['n'] for (f, d, p, c), a in agg.items() if d == dt)} comparisons."]
    open(out_md, "w").write("\n".join(out) + "\n")


def write_matrix_vjp_table(jsonl, out_md, tag):
    """tests/test_gpu_matrix_vjp.py records the worst relative error it measured per check: one row per (pullback, K, dtype)."""
    from collections import OrderedDict

    rows = [json.loads(l) for l in open(jsonl) if l.strip()]
    agg = OrderedDict()
    for r in rows:
        kind = r["what"].split(" K=")[0].replace(" no ladj", "").replace(" single", "")
        a = agg.setdefault((kind, r["K"], r["dtype"]), {"err": 0.0, "cond": 0.0, "condw": 0.0})
        if r["worst_rel_err"] >= a["err"]:
            a["err"], a["condw"] = r["worst_rel_err"], r["cond_L_at_worst"]
        a["cond"] = max(a["cond"], r["cond_L_max"])
    out = ["# Measured errors of the matrix-bijector pullbacks (`tests/test_gpu_matrix_vjp.py`, GPU vs the FD-pinned oracle)", "",
           "Worst relative error of a sample's cotangent on that sample's scale, per (kind, direction, K, dtype), over the test's batches; `cond(L)` = condition",
           "number of the triangular factor of the sample (sqrt(cond(X))): of the worst sample and the largest in the batch.  The bar is north_star's FLAT 1e-3",
           f"(Float32) / 1e-6 (Float64).  Source: the pytest run of `profiles/{tag}_pytest_gpu_tail.txt`.", "",
           "| pullback | K | dtype | worst rel. error | cond(L) of that sample | largest cond(L) in the batch |", "|---|---|---|---|---|---|"]
    for (kind, K, dt), a in sorted(agg.items(), key=lambda kv: (kv[0][0], kv[0][2], kv[0][1])):
        out.append(f"| {kind} | {K} | {dt} | {a['err']:.2e} | {a['condw']:.3g} | {a['cond']:.3g} |")
    for dt, bar in (("float32", "1e-3"), ("float64", "1e-6")):
        vals = [a["err"] for (k, K, d), a in agg.items() if d == dt]
        if vals:
            out.append("")
            out.append(f"Worst {dt}: {max(vals):.2e} (bar {bar}).")
    open(out_md, "w").write("\n".join(out) + "\n")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
