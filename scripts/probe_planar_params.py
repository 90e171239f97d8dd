#!/usr/bin/env python3
"""vjp_params(n × PlanarLayer) over column heights: the parameter reduction with rows owned by threads (planar_param_rows_kernel,
round 5) against the register accumulators (BJX_PLANAR_PARAM_ROWS=0 in the environment), and the heights the latter never served.

    python scripts/probe_planar_params.py [--log2-elems 29] [--layers 8] [--steps 5]

Per height: 2^log2-elems / dim columns; the roofline bytes are the five array passes of the call (x, ȳ read twice, x̄ written:
5·dim·sizeof(T) + the s̄ / tanh work arrays) ÷ the stream-region time of one call."""
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
    ap.add_argument("--dims", default="200,256,333,509,512,1000,1024,1500,2048,4099,8192,16384")
    a = ap.parse_args()
    torch.manual_seed(0)
    print(f"BJX_PLANAR_PARAM_ROWS={os.environ.get('BJX_PLANAR_PARAM_ROWS', '(default 1)')}")
    print("| dtype | rows | columns | ms / call | GB/s (5 passes) | % of 8 TB/s | input pullback alone, ms | its % (3 passes) | forward ms | % (2 passes) | inverse ms | % (2 passes) |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for dt in (torch.float32, torch.float64):
        es = 4 if dt == torch.float32 else 8
        for dim in [int(v) for v in a.dims.split(",")]:
            N = max(64, (1 << a.log2_elems) // (dim * (es // 4)))
            nl = a.layers
            w = torch.randn(dim, nl, device="cuda", dtype=dt) / dim ** 0.5
            u = torch.randn(dim, nl, device="cuda", dtype=dt) / dim ** 0.5
            b = torch.randn(nl, device="cuda", dtype=dt)
            layer = bj.PlanarLayer(w, u, b)
            x = torch.randn(N, dim, device="cuda", dtype=dt).T
            g = torch.randn(N, dim, device="cuda", dtype=dt).T
            lb = torch.randn(N, device="cuda", dtype=dt)
            try:
                for _ in range(2):
                    bj.vjp_params(layer, x, g, lb)
            except NotImplementedError as e:
                print(f"| {str(dt)[6:]} | {dim} | {N} | refused: {str(e)[:60]} | | |")
                continue
            k_ms, ms = kernel_and_region_ms(bj, lambda: bj.vjp_params(layer, x, g, lb), steps=a.steps, warm=2)
            _, ms_in = kernel_and_region_ms(bj, lambda: bj.vjp(layer, x, g, lb), steps=a.steps, warm=2)
            _, ms_f = kernel_and_region_ms(bj, lambda: bj.with_logabsdet_jacobian(layer, x), steps=a.steps, warm=2)
            y = bj.transform(layer, x)
            _, ms_i = kernel_and_region_ms(bj, lambda: bj.with_logabsdet_jacobian(bj.inverse(layer), y), steps=a.steps, warm=2)
            del y
            byts = (5 * dim + 4 * nl + 1) * es * N
            gbs = byts / ms / 1e6
            print(f"| {str(dt)[6:]} | {dim} | {N} | {ms:.3f} | {gbs:.0f} | {gbs / 80:.1f} | {ms_in:.3f} | {3 * dim * es * N / ms_in / 1e6 / 80:.1f} | {ms_f:.3f} | {2 * dim * es * N / ms_f / 1e6 / 80:.1f} | {ms_i:.3f} | {2 * dim * es * N / ms_i / 1e6 / 80:.1f} |")
            del x, g, lb


if __name__ == "__main__":
    main()
