#!/usr/bin/env python3
"""Float64 rows (the dtype of the reference's own tests and Turing's default): dominant-kernel time from per-launch hipEvent
pairs, algorithmic bytes (8 B per element in and out + 8 B per sample of log-det), fraction of the 8 TB/s HBM peak.

    python scripts/bench_f64.py [--log2-batch 21] [--steps 5]
"""
import argparse
import ctypes as C
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bijectors_amd as bj  # noqa: E402
from _timing import kernel_ms  # noqa: E402

f64 = torch.float64


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2-batch", type=int, default=21)
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    L, ctx = bj._lib, bj.context(dev)
    lib = L.load()

    def cm(rows, n):
        return torch.empty((n, rows), dtype=f64, device=dev).T

    def randn(rows, n, seed, std=1.0):
        t = cm(rows, n)
        L.check(ctx.h, lib.bjx_fill_normal(ctx.h, L.BJX_F64, t.data_ptr(), rows, n, 0, seed, 0.0, std), "fill")
        return t

    N = 1 << a.log2_batch
    d = 64
    rows = []
    x = randn(d, N, 0)
    e = bj.elementwise
    sh = bj.shard.with_logabsdet_jacobian_sharded
    y = cm(d, N)
    c2 = e(bj.exp) @ bj.Shift(0.1) @ bj.Scale(0.5)
    rows.append(("exp∘Shift∘Scale d=64 (sum)", lambda: sh(c2, x, out=y, per_sample=False), 16 * d, N))
    lg = bj.Logit(0.0, 1.0)
    xu = torch.sigmoid(x.T).T
    rows.append(("Logit(0,1) d=64", lambda: sh(lg, xu, out=y, per_sample=False), 16 * d, N))
    rows.append(("OrderedBijector d=64", lambda: sh(bj.OrderedBijector(), x, out=y), 16 * d + 8, N))
    Ks = 32
    xs = torch.softmax(randn(Ks, N, 1).T, dim=1).T
    rows.append(("SimplexBijector K=32", lambda: sh(bj.SimplexBijector(), xs), 8 * (2 * Ks - 1) + 8, N))
    ys = randn(Ks - 1, N, 2)
    rows.append(("inverse(SimplexBijector) K=32", lambda: sh(bj.inverse(bj.SimplexBijector()), ys), 8 * (2 * Ks - 1) + 8, N))
    K = 64
    Nc = 1 << min(a.log2_batch, 15)
    n = K * (K - 1) // 2
    yv = randn(n, Nc, 3, std=0.5)
    icb = bj.inverse(bj.VecCholeskyBijector("U"))
    rows.append(("inverse(VecCholesky) K=64", lambda: sh(icb, yv), 8 * (n + K * K) + 8, Nc))
    Wd = sh(icb, yv)[0]
    cb = bj.VecCholeskyBijector("U")
    rows.append(("VecCholesky K=64 (W→y)", lambda: sh(cb, Wd), 8 * (n + K * K) + 8, Nc))
    dp, nl = 128, 8
    Np = 1 << min(a.log2_batch, 20)
    z = randn(dp, Np, 4)
    w = randn(dp, nl, 5, std=1 / math.sqrt(dp))
    u = randn(dp, nl, 6, std=1 / math.sqrt(dp))
    bb = randn(nl, 1, 7).reshape(-1).contiguous()
    flow = bj.PlanarLayer(w, u, bb)
    yz = cm(dp, Np)
    rows.append(("8×PlanarLayer d=128", lambda: sh(flow, z, out=yz), 16 * dp + 8, Np))
    zf = bj.transform(flow, z)
    rows.append(("inverse(8×PlanarLayer) d=128", lambda: sh(bj.inverse(flow), zf, out=yz), 16 * dp + 8, Np))
    # the pullbacks (round 5: planar_vjp_cols_kernel / planar_param_rows_kernel serve Float64 — the lanes-per-column kernel ran them at 14 %)
    gz = randn(dp, Np, 11)
    lbz = randn(Np, 1, 12).reshape(-1).contiguous()
    rows.append(("vjp(8×PlanarLayer) d=128", lambda: bj.vjp(flow, z, gz, lbz), 24 * dp + 8, Np))
    rows.append(("vjp(inverse(8×PlanarLayer)) d=128", lambda: bj.vjp(bj.inverse(flow), zf, gz, lbz), 24 * dp + 8, Np))
    rows.append(("vjp_params(8×PlanarLayer) d=128 (two passes)", lambda: bj.vjp_params(flow, z, gz, lbz), 40 * dp + 8 + 16 * nl, Np))
    dr, Kb = 32, 16
    raw = [randn(dr, k, 100 + i) for i, k in enumerate((Kb, Kb, Kb - 1))]
    rqs = bj.RationalQuadraticSpline(raw[0], raw[1], raw[2], 3.0)
    xr = randn(dr, N, 9)
    yr = cm(dr, N)
    rows.append(("RQS K=16 d=32 forward", lambda: sh(rqs, xr, out=yr), 16 * dr + 8, N))
    print("| row (Float64) | kernel ms | samples | alg. B/sample | GB/s | % of 8 TB/s |")
    print("|---|---|---|---|---|---|")
    for name, step, bps, ns in rows:
        try:
            k = kernel_ms(bj, step, steps=a.steps, device=dev)
            gbs = bps * ns / (k * 1e-3) / 1e9
            print(f"| {name} | {k:.4f} | 2^{int(math.log2(ns))} | {bps} | {gbs:.0f} | {100 * gbs / 8000:.1f} |", flush=True)
        except Exception as ex:
            print(f"| {name} | failed: {ex!r} | | | | |", flush=True)


if __name__ == "__main__":
    main()
