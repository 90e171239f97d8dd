"""Every BJX_* tuning switch the library reads (`getenv` in csrc/) selects another kernel or another tile geometry for some shapes.
They exist for same-box A/Bs, but each one is a code path a user can reach, so each one is run here: a worker process executes the
same battery of calls under the switch, and every result must agree with the default process's (north_star tolerances — the kernels
differ in summation order, not in what they compute).  One process per setting (the switches are read once per process)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SETTINGS = [
    "BJX_NT=0", "BJX_CHAIN_WALKER=0", "BJX_CHAIN_FLATCOL=0", "BJX_CHAIN_COLBATCH=0", "BJX_COLWALK=0", "BJX_PLANAR_REG=0", "BJX_PLANAR_SPLIT=0",
    "BJX_PLANAR_SPLIT=1", "BJX_PLANAR_MFMA=1", "BJX_PLANAR_MFMA=2", "BJX_PLANAR_MFMA=4", "BJX_PLANAR_MFMA64=0", "BJX_PLANAR_TILE=0",
    "BJX_FLOW_WALK_MAX=0", "BJX_RADIAL_WALK_ALL=1", "BJX_PLANAR_PARAM_MFMA=0", "BJX_SCALE_MFMA=0", "BJX_MATRIX_LANE_MAX=0",
    "BJX_MATRIX_LANE_DIRECT=0", "BJX_MATRIX_CYC=0", "BJX_MATRIX_VJP_GRP=0", "BJX_MATRIX_VJP_MFMA=0", "BJX_MATRIX_VJP_MFMA=2", "BJX_SEQ_WAVE=0", "BJX_SEQ_STREAM=0", "BJX_ORDERED_VJP_STREAM=0", "BJX_SIMPLEX_VJP_STREAM=0",
    "BJX_SEQ_TINY=0", "BJX_SEQ_TALL=0", "BJX_SIMPLEX_VJP_TALL=0", "BJX_CHAIN_TINY=0", "BJX_PLANAR_REG_UNALIGNED=0", "BJX_PLANAR_REG_BIG=0",
    "BJX_FLOW_UNALIGNED=0", "BJX_COL_UNALIGNED=0", "BJX_STACKED_VJP_UNALIGNED=0", "BJX_COL_SLAB=0", "BJX_STACKED_SLAB=0", "BJX_STACKED_SLAB=64",
    "BJX_CHAIN_UNALIGNED=0", "BJX_PLANAR_WALK_DIRECT=0", "BJX_COLDIRECT=0", "BJX_STACKED_TINY=0", "BJX_CHOL_CHUNK=0", "BJX_CHOL_LANE_MAX=0",
    "BJX_CHOL_FWD_VJP_SWZ=0", "BJX_STACKED_WALKER=0", "BJX_RQS_SLAB=0", "BJX_RQS_SLAB=32", "BJX_RQS_KNOTS_SHARED=0", "BJX_RQS_KNOTS_SHARED=2", "BJX_SCALE_PREP_WAVE=0", "BJX_ORDERED_VJP_TALL=0",
    "BJX_PLANAR_COLS_MIN_F32=0", "BJX_PLANAR_COLS_MIN_F64=0", "BJX_PLANAR_VJP_COLS_MIN_F32=0", "BJX_PLANAR_VJP_COLS_MIN_F64=0", "BJX_PLANAR_PARAM_ROWS=0", "BJX_PLANAR_PARAM_ROWS=33", "BJX_PLANAR_TILE_MAX_F64=128", "BJX_PLANAR_TILE_MAX_F64=32",
]


def _run(tmp, name, env_extra):
    path = os.path.join(tmp, name + ".npz")
    env = dict(os.environ)
    env.update(env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_env_switch_worker.py"), path], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, f"{env_extra}: worker failed\n{p.stdout[-2000:]}\n{p.stderr[-3000:]}"
    return dict(np.load(path))


@pytest.fixture(scope="module")
def reference(tmp_path_factory):
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    tmp = str(tmp_path_factory.mktemp("envsw"))
    return tmp, _run(tmp, "default", {})


def test_the_list_covers_every_switch_in_the_sources():
    names = set()
    for f in os.listdir(os.path.join(ROOT, "bijectors.jl_amd", "csrc")):
        if f.endswith((".hip", ".h")):
            src = open(os.path.join(ROOT, "bijectors.jl_amd", "csrc", f)).read()
            names |= set(re.findall(r'(?:getenv|env_int)\("(BJX_[A-Z0-9_]+)"', src))
    covered = {s.split("=")[0] for s in SETTINGS}
    assert names == covered, f"switches without a test: {sorted(names - covered)}; listed but gone: {sorted(covered - names)}"


@pytest.mark.parametrize("setting", SETTINGS)
def test_switch_gives_the_same_results(reference, setting):
    tmp, ref = reference
    k, v = setting.split("=")
    got = _run(tmp, setting.replace("=", "_"), {k: v})
    assert set(got) == set(ref)
    for name in sorted(ref):
        a, b = got[name], ref[name]
        f32 = ".f32" in name
        scale = max(1.0, float(np.abs(b).max()) if b.size else 1.0)
        tol = (2e-3 if f32 else 1e-6)
        assert a.shape == b.shape, name
        if not np.allclose(a, b, rtol=tol, atol=tol * scale * (8 if "vjp" in name or "chol" in name or "vcorr" in name else 1), equal_nan=True):
            worst = float(np.nanmax(np.abs(a - b)))
            raise AssertionError(f"{setting}: {name} differs from the default path by {worst:.3g} (scale {scale:.3g})")
