#!/usr/bin/env python3
"""GPU time of SMALL calls (f-2 shapes: param_dim x n_chains, src/vector/product/fill.jl:146-165): stream-region time per call over a burst
of 200 calls issued through cached launch plans (host ~9 us per call), for heights that are / are not whole 16-byte packs.
   python scripts/probe_small_calls.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bijectors_amd as bj  # noqa: E402


def region_us(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        host = (time.perf_counter() - t0) / n * 1e6
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best, host


def main():
    dev = torch.device("cuda", 0)
    V = bj.vector
    print("| bijector | rows x chains | dtype | log-det | stream us / call | host issue us / call |")
    print("|---|---|---|---|---|---|")
    for dt in (torch.float32, torch.float64):
        for rows, chains in ((1000, 16), (1001, 16), (999, 16), (64, 256), (63, 256), (65, 256), (10, 1000), (7, 4096), (256, 64), (4096, 8)):
            x = torch.randn(chains, rows, device=dev, dtype=dt).T
            pos = V.from_linked_vec(V.scalar_to_scalar_bijector(0.0, float("inf")), (rows,))
            unit = V.from_linked_vec(V.scalar_to_scalar_bijector(0.0, 1.0), (rows,))
            chain = bj.elementwise(bj.exp) @ bj.Shift(0.1) @ bj.Scale(0.5)
            for name, b, ps in (("from_linked_vec(positive)", pos, True), ("from_linked_vec(unit interval)", unit, True), ("exp∘Shift∘Scale", chain, False), ("exp∘Shift∘Scale", chain, True)):
                us, host = region_us(lambda: bj.with_logabsdet_jacobian(b, x, per_sample=ps))
                print(f"| {name} | {rows} x {chains} | {str(dt)[6:]} | {'per chain' if ps else 'scalar'} | {us:.1f} | {host:.1f} |")


if __name__ == "__main__":
    main()
