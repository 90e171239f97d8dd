#!/usr/bin/env python3
"""vjp_params(n × PlanarLayer) over column heights: the parameter reduction with rows owned by threads (planar_param_rows_kernel,
round 5) against the register accumulators (BJX_PLANAR_PARAM_ROWS=0 in the environment), and the heights the latter never served.

    python scripts/probe_planar_params.py [--log2-elems 29] [--layers 8] [--steps 5]

Per height: 2^log2-elems / dim columns (half of them in Float64); per operation the stream-region time of one call and the algorithmic
array passes (vjp_params: x, ȳ read twice, x̄ written = 5; vjp 3; forward / inverse 2) as a fraction of 8 TB/s."""
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
    ap.add_argument("--log2-elems", type=int, default=29)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--kind", default="planar", choices=["planar", "radial"])
    ap.add_argument("--dims", default="64,128,200,256,333,512,1000,1024,1500,2048,4099,8192,16384")
    a = ap.parse_args()
    torch.manual_seed(0)
    print("| dtype | rows | columns | vjp_params ms | % of 8 TB/s (5 passes) | vjp ms | % (3 passes) | forward ms | % (2 passes) | inverse ms | % (2 passes) |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for dt in (torch.float32, torch.float64):
        es = 4 if dt == torch.float32 else 8
        for dim in [int(v) for v in a.dims.split(",")]:
            N = max(64, (1 << a.log2_elems) // (dim * (es // 4)))
            nl = a.layers
            w = torch.randn(dim, nl, device="cuda", dtype=dt) / dim ** 0.5
            u = torch.randn(dim, nl, device="cuda", dtype=dt) / dim ** 0.5
            b = torch.randn(nl, device="cuda", dtype=dt)
            layer = bj.PlanarLayer(w, u, b) if a.kind == "planar" else bj.RadialLayer(torch.tensor([0.3], device="cuda", dtype=dt), torch.tensor([0.7], device="cuda", dtype=dt), w[:, 0].contiguous())
            x = torch.randn(N, dim, device="cuda", dtype=dt).T
            g = torch.randn(N, dim, device="cuda", dtype=dt).T
            lb = torch.randn(N, device="cuda", dtype=dt)
            def cell(fn, passes):
                try:
                    fn()
                    _, ms = kernel_and_region_ms(bj, fn, steps=a.steps, warm=2)
                    return f"{ms:.3f} | {passes * dim * es * N / ms / 1e6 / 80:.1f}"
                except NotImplementedError:
                    return "refused | —"

            y = bj.transform(layer, x)
            cells = [cell(lambda: bj.vjp_params(layer, x, g, lb), 5), cell(lambda: bj.vjp(layer, x, g, lb), 3),
                     cell(lambda: bj.with_logabsdet_jacobian(layer, x), 2), cell(lambda: bj.with_logabsdet_jacobian(bj.inverse(layer), y), 2)]
            print(f"| {str(dt)[6:]} | {dim} | {N} | " + " | ".join(cells) + " |", flush=True)
            del y
            del x, g, lb


if __name__ == "__main__":
    main()
