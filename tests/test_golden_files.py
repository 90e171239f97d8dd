"""Consumes tests/golden/*.json (scripts/gen_golden.py + scripts/gen_golden.jl): every file holds inputs and the outputs of
`with_logabsdet_jacobian`.  With "source": "Bijectors.jl …" the outputs come from the real package and PIN the oracle
(rtol 1e-9 — the reference's own link/invlink tolerance, test/legacy_interface.jl:59-66); with "source": "oracle" (the
state of this repository: Julia is not in the image) the test only proves that the files, the case registry and the
consumer agree, so that running the Julia script upgrades the pin without touching the tests."""
import glob
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from golden_cases import cases  # noqa: E402

FILES = sorted(glob.glob(os.path.join(HERE, "golden", "*.json")))


def _arr(v):
    a = np.asarray(v, dtype=np.float64)
    return a.T if a.ndim == 2 else a          # stored as a list of columns


def test_every_case_has_a_file():
    assert {os.path.basename(f)[:-5] for f in FILES} == set(cases()), "run scripts/gen_golden.py"


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-5] for f in FILES])
def test_oracle_reproduces_the_golden_file(orc, path):
    doc = json.load(open(path))
    c = cases()[doc["name"]]
    assert doc["julia"] == c["julia"]
    pinned = doc["source"].startswith("Bijectors.jl")
    rtol, atol = (1e-9, 1e-11) if pinned else (1e-13, 1e-14)
    for x, y_ref, l_ref in zip(doc["x"], doc["y"], doc["logabsdetjac"]):
        y, l = c["fn"](doc["params"], x)
        np.testing.assert_allclose(np.asarray(y, dtype=np.float64), _arr(y_ref), rtol=rtol, atol=atol, err_msg=doc["name"])
        assert abs(float(np.asarray(l).reshape(-1)[0]) - l_ref) <= rtol * abs(l_ref) + atol * 10, doc["name"]
