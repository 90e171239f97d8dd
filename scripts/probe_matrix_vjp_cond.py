"""How far the two forward pullbacks of PDVecBijector (group kernel: two triangular solves; MFMA kernel: explicit block inverse) are
from the Float64 oracle on an ILL-conditioned factor (the K = 64 case the env-switch battery first used: off-diagonals 0.4 N(0,1),
cotangent scale 2e6).  Run once per switch value:  BJX_MATRIX_VJP_MFMA=0|1 python scripts/probe_matrix_vjp_cond.py"""
import os
import sys

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "oracle"))
import bijectors_amd as bj  # noqa: E402
import oracle as orc  # noqa: E402

r = np.random.default_rng(3)
for K, N, sc in ((64, 9, 0.4), (64, 9, 0.1), (64, 9, 0.025), (32, 9, 0.4), (32, 9, 0.1)):
    pv = bj.PDVecBijector()
    ypd = np.asfortranarray((sc * r.normal(size=(K * (K + 1) // 2, N))))
    lb = r.normal(size=N)
    gy = np.asfortranarray(r.normal(size=ypd.shape))
    X64 = orc.matrix_bijector("pd_vec", ypd, inverse=True)[0] if hasattr(orc, "matrix_bijector") else None
    dev = lambda a, dt: torch.from_numpy(np.ascontiguousarray(np.moveaxis(a, -1, 0))).to(dt).cuda().movedim(0, -1)
    Xd = bj.transform(bj.inverse(pv), dev(ypd, torch.float64))
    X = Xd.cpu().numpy()
    ref = orc.matrix_bijector_vjp("pd_vec", X, gy, lb, inverse=False)
    got = bj.vjp(pv, Xd.to(torch.float32), dev(gy, torch.float32), torch.from_numpy(lb).float().cuda()).cpu().numpy().astype(np.float64)
    scale = np.abs(ref).reshape(-1, N).max(axis=0)
    err = np.abs(got - ref).reshape(-1, N).max(axis=0) / scale
    condL = [np.linalg.cond(np.linalg.cholesky(X[:, :, n])) for n in range(N)]
    print(f"K={K} offdiag {sc}: worst rel err of a sample {err.max():.3g} (median {np.median(err):.3g}), cotangent scale {scale.max():.3g}, cond(L) {min(condL):.3g} .. {max(condL):.3g}  [MFMA={os.environ.get('BJX_MATRIX_VJP_MFMA', '1')}]")
