#!/usr/bin/env python3
"""Ordered pullback on columns beyond the stream kernel (2 048 rows Float32): kernel and stream-region fraction of the HBM peak."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

import bijectors_amd as bj  # noqa: E402
from _timing import kernel_and_region_ms  # noqa: E402

dev = torch.device("cuda", 0)
print("| pullback | rows x columns | kernel ms | % of 8 TB/s | % by stream region |")
print("|---|---|---|---|---|")
for rows, N in ((2048, 1 << 16), (4096, 1 << 15), (5003, 1 << 14), (16384, 1 << 13)):
    y = (0.3 * torch.randn(N, rows, device=dev)).T
    g = torch.randn(N, rows, device=dev).T
    lb = torch.randn(N, device=dev)
    for inv in (False, True):
        b = bj.inverse(bj.OrderedBijector()) if inv else bj.OrderedBijector()
        x = bj.transform(bj.OrderedBijector(), y) if inv else y
        k, r = kernel_and_region_ms(bj, lambda: bj.vjp(b, x, g, lb), steps=10, device=dev)
        byts = (3 * rows * 4 + 4) * N
        name = "vjp(inverse(OrderedBijector))" if inv else "vjp(OrderedBijector)"
        print(f"| {name} | {rows} x 2^{N.bit_length() - 1} | {k:.4f} | {byts / k / 1e6 / 8000 * 100:.1f} | {byts / r / 1e6 / 8000 * 100:.1f} |")
