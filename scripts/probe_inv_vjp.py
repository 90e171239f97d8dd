#!/usr/bin/env python3
"""vjp(inverse(8 x PlanarLayer)) over column heights, Float32 and Float64: stream-region time of one call and its three array
passes as a fraction of 8 TB/s.  With BJX_LIB_PATH pointing at another build of the library this is a same-box A/B (round 5: the
safeguarded find_alpha loop in the pullback kernels against the fast root solve)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch  # noqa: E402

import bijectors_amd as bj  # noqa: E402
from _timing import kernel_and_region_ms  # noqa: E402

for dt in (torch.float32, torch.float64):
    es = 4 if dt == torch.float32 else 8
    for dim in (4, 8, 16, 32, 128, 200, 512, 1500):
        N = (1 << 29) // (dim * (es // 4))
        nl = 8
        w = torch.randn(dim, nl, device="cuda", dtype=dt) / dim ** 0.5
        u = torch.randn(dim, nl, device="cuda", dtype=dt) / dim ** 0.5
        layer = bj.PlanarLayer(w, u, torch.randn(nl, device="cuda", dtype=dt))
        x = torch.randn(N, dim, device="cuda", dtype=dt).T
        g = torch.randn(N, dim, device="cuda", dtype=dt).T
        lb = torch.randn(N, device="cuda", dtype=dt)
        y = bj.transform(layer, x)
        _, ms = kernel_and_region_ms(bj, lambda: bj.vjp(bj.inverse(layer), y, g, lb), steps=5, warm=2)
        print(f"{str(dt)[6:]} {dim}: vjp(inverse) {ms:.3f} ms  {3 * dim * es * N / ms / 1e6 / 80:.1f} %", flush=True)
        del x, g, lb, y
