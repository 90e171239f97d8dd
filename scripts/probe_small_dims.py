#!/usr/bin/env python3
"""Low-dimensional columns (the reference's everyday shapes: 2 ... 50 rows), Float32 and Float64, every hot-path family and its
pullback: stream-region time of one call and the algorithmic array passes as a fraction of 8 TB/s.  Looks for paths that fell
behind (round 5 found the Float64 inverse Planar pullback on a safeguarded Float64 loop: 3-6 %).

    python scripts/probe_small_dims.py [--log2-elems 27]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch  # noqa: E402

import bijectors_amd as bj  # noqa: E402
from _timing import kernel_and_region_ms  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2-elems", type=int, default=27)
    ap.add_argument("--dims", default="2,5,10,20,50")
    a = ap.parse_args()
    torch.manual_seed(0)
    print("| dtype | rows | operation | ms | % of 8 TB/s |")
    print("|---|---|---|---|---|")
    for dt in (torch.float32, torch.float64):
        es = 4 if dt == torch.float32 else 8
        for dim in [int(v) for v in a.dims.split(",")]:
            N = (1 << a.log2_elems) // dim
            x = torch.randn(N, dim, device="cuda", dtype=dt).T
            g = torch.randn(N, dim, device="cuda", dtype=dt).T
            lb = torch.randn(N, device="cuda", dtype=dt)
            xs = torch.softmax(torch.randn(N, dim, device="cuda", dtype=dt), dim=1).T if dim >= 2 else None
            gs = torch.randn(N, dim - 1, device="cuda", dtype=dt).T if dim >= 2 else None      # the Simplex link has K - 1 rows
            nl = 8
            w = torch.randn(dim, nl, device="cuda", dtype=dt) / dim ** 0.5
            u = torch.randn(dim, nl, device="cuda", dtype=dt) / dim ** 0.5
            pl = bj.PlanarLayer(w, u, torch.randn(nl, device="cuda", dtype=dt))
            rd = bj.RadialLayer(torch.tensor([0.3], device="cuda", dtype=dt), torch.tensor([0.7], device="cuda", dtype=dt), w[:, 0].contiguous())
            ch = bj.elementwise(bj.exp) @ bj.Shift(0.1) @ bj.Scale(torch.linspace(0.5, 1.5, dim, device="cuda", dtype=dt))
            y_pl = bj.transform(pl, x)
            y_rd = bj.transform(rd, x)
            ops = [("chain exp∘Shift∘Scale(v) per-sample", lambda: bj.with_logabsdet_jacobian(ch, x, per_sample=True), 2),
                   ("vjp(chain)", lambda: bj.vjp(ch, x, g, lb), 3),
                   ("Ordered", lambda: bj.with_logabsdet_jacobian(bj.OrderedBijector(), x, per_sample=True), 2),
                   ("vjp(Ordered)", lambda: bj.vjp(bj.OrderedBijector(), x, g, lb), 3),
                   ("Simplex", (lambda: bj.with_logabsdet_jacobian(bj.SimplexBijector(), xs, per_sample=True)) if dim >= 2 else None, 2),
                   ("vjp(Simplex)", (lambda: bj.vjp(bj.SimplexBijector(), xs, gs, lb)) if dim >= 2 else None, 3),
                   ("8×Planar", lambda: bj.with_logabsdet_jacobian(pl, x), 2),
                   ("inverse(8×Planar)", lambda: bj.with_logabsdet_jacobian(bj.inverse(pl), y_pl), 2),
                   ("vjp(8×Planar)", lambda: bj.vjp(pl, x, g, lb), 3),
                   ("vjp(inverse(8×Planar))", lambda: bj.vjp(bj.inverse(pl), y_pl, g, lb), 3),
                   ("vjp_params(8×Planar)", lambda: bj.vjp_params(pl, x, g, lb), 5),
                   ("Radial", lambda: bj.with_logabsdet_jacobian(rd, x), 2),
                   ("inverse(Radial)", lambda: bj.with_logabsdet_jacobian(bj.inverse(rd), y_rd), 2),
                   ("vjp(Radial)", lambda: bj.vjp(rd, x, g, lb), 3),
                   ("vjp(inverse(Radial))", lambda: bj.vjp(bj.inverse(rd), y_rd, g, lb), 3),
                   ("vjp_params(Radial)", lambda: bj.vjp_params(rd, x, g, lb), 3)]
            for name, fn, passes in ops:
                if fn is None:
                    continue
                try:
                    fn()
                    _, ms = kernel_and_region_ms(bj, fn, steps=5, warm=2)
                    print(f"| {str(dt)[6:]} | {dim} | {name} | {ms:.3f} | {passes * dim * es * N / ms / 1e6 / 80:.1f} |", flush=True)
                except Exception as e:  # noqa: BLE001
                    print(f"| {str(dt)[6:]} | {dim} | {name} | failed: {str(e)[:60]} | |", flush=True)
            del x, g, lb, xs, gs, y_pl, y_rd


if __name__ == "__main__":
    main()
