#!/usr/bin/env python3
"""Median / min / mean of the dominant kernel's launches in the TIMED part of a rocprofv3 kernel trace (VERDICT r05 weak #7a: the
`--stats` mean contains the cold launches of warm-up and pre-roll, so the line's `frac` did not follow from profiles/).

  trace_stats.py <kernel_trace.csv> <workload> [last_n]

bench.py runs, per workload: warm-up, a cold pass, the pre-roll (≥ 60 ms of the same step), then THREE timed passes of `steps` steps.
The last `last_n` launches of every distinct dominant kernel (default 60 = 3 passes × 20 steps) are those timed passes."""
import csv
import json
import statistics
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from collect_profiles import DOMINANT  # noqa: E402


def main(path, wl, last_n=60):
    subs = DOMINANT[wl]
    by = {}
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if any(s in name for s in subs):
            by.setdefault(name, []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    out = {"workload": wl, "kernels": []}
    for name, v in by.items():
        v.sort()
        d = [x[1] / 1e3 for x in v]              # µs
        t = d[-last_n:]
        out["kernels"].append({"kernel": name[:140], "launches_in_trace": len(d), "timed_launches": len(t),
                               "timed_median_us": round(statistics.median(t), 2), "timed_min_us": round(min(t), 2), "timed_mean_us": round(sum(t) / len(t), 2),
                               "timed_max_us": round(max(t), 2), "all_mean_us": round(sum(d) / len(d), 2), "all_max_us": round(max(d), 2)})
    out["sum_of_timed_medians_us"] = round(sum(k["timed_median_us"] for k in out["kernels"]), 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 60)
