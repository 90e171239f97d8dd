#!/usr/bin/env python3
"""profiles/<TAG>_pmc_notes.md from the SQ-counter summaries of `scripts/gpu_run.sh profile <TAG>` (<TAG>_c3_sq_pmc_{1,2,3}.txt, <TAG>_c4_f64_sq_pmc_{1,2,3}.txt):
per-element issue slots, VALU busy, the split of the wave cycles, LDS conflicts.   python scripts/make_pmc_notes.py r06"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse(fn):
    out, cur = {}, None
    for l in open(fn):
        if l.startswith("kernel:"):
            m = re.search(r"(rqs_lds_kernel<float, 4, 5, true, (true|false)>|planar_mfma64_kernel<[^>]*>)", l)
            cur = m.group(1) if m else None
            if cur:
                out.setdefault(cur, {})
        elif cur:
            p = l.split()
            if len(p) >= 2:
                out[cur][p[0]] = float(p[1])
    return out


def main(tag):
    P = lambda n: os.path.join(ROOT, "profiles", f"{tag}_{n}")
    c3, c4 = {}, {}
    for i in (1, 2, 3):
        for k, v in parse(P(f"c3_sq_pmc_{i}.txt")).items():
            c3.setdefault(k, {}).update(v)
        for k, v in parse(P(f"c4_f64_sq_pmc_{i}.txt")).items():
            c4.setdefault(k, {}).update(v)
    el = 32 * (1 << 22) / 64.0
    L = [f"# {tag} — issue-side counters of the two BASELINE kernels under their roofline, ON THE SHIPPED LIBRARY (VERDICT r05 missing #7)", "",
         f"Collected by `scripts/gpu_run.sh profile {tag}` (rocprofv3 `--pmc` + `--kernel-trace` only, one pass per counter set, `bench.py --workload <wl> --steps 2",
         f"--warmup 1`) with the library of `profiles/{tag}_lib_sha16.txt`; per-kernel means in `{tag}_c3_sq_pmc_{{1,2,3}}.txt`, `{tag}_c4_f64_sq_pmc_{{1,2,3}}.txt`; this file is",
         "`scripts/make_pmc_notes.py` over them.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md); chip totals; GRBM_GUI_ACTIVE is summed over the 8 XCDs.", "",
         "## C3 — `rqs_lds_kernel<float, 4, 5, true, INV>`, 4 520 blocks × 256 threads, 2²² columns × 32 rows (the Float32 kernel is round 5's: the round's",
         f"experiments were measured and reverted, `profiles/{tag}_c3_experiments.md`)", "", "| | forward | inverse |", "|---|---|---|"]
    f, v = c3["rqs_lds_kernel<float, 4, 5, true, false>"], c3["rqs_lds_kernel<float, 4, 5, true, true>"]
    row = lambda name, fn: L.append(f"| {name} | {fn(f)} | {fn(v)} |")
    row("SQ_INSTS_VALU per 64 elements", lambda d: f"{d['SQ_INSTS_VALU'] / el:.1f}")
    row("SQ_ACTIVE_INST_VALU per 64 elements (issue slots; a transcendental holds the pipe for 4)", lambda d: f"{d['SQ_ACTIVE_INST_VALU'] / el:.1f}")
    row("SQ_INSTS_LDS / SQ_INSTS_SALU per 64 elements", lambda d: f"{d['SQ_INSTS_LDS'] / el:.1f} / {d['SQ_INSTS_SALU'] / el:.1f}")
    row("VALU busy = ACTIVE_INST_VALU ÷ (GRBM_GUI_ACTIVE/8/4 × 1 024 SIMDs)", lambda d: f"**{100 * d['SQ_ACTIVE_INST_VALU'] / (d['GRBM_GUI_ACTIVE'] / 32 * 1024):.0f} %**")
    row("wave cycles: issuing / issue-stalled (WAIT_INST_ANY) / parked in s_waitcnt (WAIT_ANY)",
        lambda d: f"{100 * d['SQ_ACTIVE_INST_ANY'] / d['SQ_WAVE_CYCLES']:.0f} / {100 * d['SQ_WAIT_INST_ANY'] / d['SQ_WAVE_CYCLES']:.0f} / {100 * d['SQ_WAIT_ANY'] / d['SQ_WAVE_CYCLES']:.0f} %")
    row("SQ_WAIT_INST_LDS ÷ WAVE_CYCLES", lambda d: f"{100 * d['SQ_WAIT_INST_LDS'] / d['SQ_WAVE_CYCLES']:.1f} %")
    row("SQ_LDS_BANK_CONFLICT ÷ SQ_LDS_IDX_ACTIVE", lambda d: f"{100 * d['SQ_LDS_BANK_CONFLICT'] / d['SQ_LDS_IDX_ACTIVE']:.0f} %")
    row("GRBM_GUI_ACTIVE / 8 (busy cycles of the kernel; µs if the clock were 2.4 GHz)", lambda d: f"{d['GRBM_GUI_ACTIVE'] / 8 / 1e3:.0f} k ({d['GRBM_GUI_ACTIVE'] / 8 / 2.4e3:.0f} µs)")
    L += ["", "Same picture as round 5 (the kernel is the same): the inverse issues VALU ≈ 88-91 % of the time at 54 slots per element, the forward ≈ 73 % at 43 slots.",
          f"What the round tried against it and what each was worth: `profiles/{tag}_c3_experiments.md`.", "",
          "## C4 Float64 — `planar_mfma64_kernel<8, 2, false, 1>`, 16 384 blocks × 256 threads, 2²¹ columns × 128 rows, 8 layers (round 6: staging pitch RC + 2, the log-det",
          "logs off the serial recurrence)", "", "| quantity | round 6 | round 5 |", "|---|---|---|"]
    d = list(c4.values())[0]
    cyc = d["GRBM_GUI_ACTIVE"] / 8
    L.append(f"| wave cycles: issuing / issue-stalled / parked in s_waitcnt | {100 * d['SQ_ACTIVE_INST_ANY'] / d['SQ_WAVE_CYCLES']:.0f} / {100 * d['SQ_WAIT_INST_ANY'] / d['SQ_WAVE_CYCLES']:.0f} / {100 * d['SQ_WAIT_ANY'] / d['SQ_WAVE_CYCLES']:.0f} % | 19 / 33 / 48 % |")
    L.append(f"| mean resident waves per CU (WAVE_CYCLES ÷ kernel quad-cycles ÷ 256) | {d['SQ_WAVE_CYCLES'] / (cyc / 4) / 256 / 4 * 4:.1f} | 7.3 |")
    L.append(f"| VALU busy | {100 * d['SQ_ACTIVE_INST_VALU'] / (d['GRBM_GUI_ACTIVE'] / 32 * 1024):.0f} % | 20 % |")
    L.append(f"| LDS busy (SQ_LDS_IDX_ACTIVE ÷ CU cycles) | {100 * d['SQ_LDS_IDX_ACTIVE'] / 256 / cyc:.0f} % | 32 % |")
    L.append(f"| of which bank conflicts | {100 * d['SQ_LDS_BANK_CONFLICT'] / d['SQ_LDS_IDX_ACTIVE']:.0f} % | 64 % |")
    L.append(f"| SQ_WAIT_INST_LDS ÷ WAVE_CYCLES | {100 * d['SQ_WAIT_INST_LDS'] / d['SQ_WAVE_CYCLES']:.0f} % | 7 % |")
    L.append(f"| SQ_INSTS_VALU / SALU / LDS / VMEM_RD per wave | {d['SQ_INSTS_VALU'] / d['SQ_WAVES']:.0f} / {d['SQ_INSTS_SALU'] / d['SQ_WAVES']:.0f} / {d['SQ_INSTS_LDS'] / d['SQ_WAVES']:.0f} / {d['SQ_INSTS_VMEM_RD'] / d['SQ_WAVES']:.0f} | — |")
    L += ["", "Reading: the LDS side got cheaper (busy and conflicts above) and the time did not move (+2 % in the same-box A/B): the wave is parked on MEMORY (each wave loads ONE",
          "32 KiB tile, computes, stores and exits: with two waves per SIMD the load latency and the block turn-over are exposed), not on LDS and not on the recurrence (taking the",
          "log-det logs off it changed nothing).  Loading the tile DIRECTLY in the matrix-core layout — no LDS at all — was built too: parity-green and 22 % slower, because a load",
          "instruction then touches sixteen columns × 64 bytes instead of one column × 1 KiB (LAB_NOTEBOOK.md).  What is left is a persistent, double-buffered form with the coalesced",
          "loads kept, which needs the tile in fewer registers than the 128 it takes — the two-waves-per-tile design DESIGN.md §9 names.  Not built in round 6."]
    open(P("pmc_notes.md"), "w").write("\n".join(L) + "\n")
    print("\n".join(L[9:19] + L[-14:-6]))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r06")
