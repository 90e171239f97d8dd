#!/usr/bin/env python3
"""Writes tests/golden/<name>.json for every case of tests/golden_cases.py: the stored inputs and the ORACLE's outputs
("source": "oracle").  `julia scripts/gen_golden.jl` then replaces the outputs by Bijectors.jl's own and sets
"source": "Bijectors.jl <version>" — from then on tests/test_golden_files.py pins the oracle to the real package."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_cases import cases  # noqa: E402


def tolist(v):
    a = np.asarray(v, dtype=np.float64)
    if a.ndim == 2:
        return [list(map(float, c)) for c in a.T]          # list of columns (Julia: reduce(hcat, v))
    return list(map(float, a.reshape(-1))) if a.ndim else float(a)


def main():
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    for name, c in cases().items():
        ys, ls = [], []
        for x in c["xs"]:
            y, l = c["fn"](c["params"], x)
            ys.append(tolist(y))
            ls.append(float(np.asarray(l).reshape(-1)[0]))
        doc = {"name": name, "julia": c["julia"], "source": "oracle", "params": c["params"], "x": c["xs"], "y": ys, "logabsdetjac": ls}
        with open(os.path.join(out, name + ".json"), "w") as f:
            json.dump(doc, f, indent=1)
    print(f"wrote {len(cases())} files to {out}")


if __name__ == "__main__":
    main()
