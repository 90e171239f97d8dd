"""Chains with per-row vector parameters (Shift(mu), Scale(sigma)) and the SUMMED log-det at heights that are / are not whole packs."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import bijectors_amd as bj
from _timing import kernel_ms
dev = torch.device("cuda", 0)
e = bj.elementwise
for d in [int(v) for v in sys.argv[1:]]:
    N = (1 << 28) // (d * 4) // 64 * 64
    x = torch.randn(N, d, device=dev).T
    mu = torch.randn(d, device=dev); sg = torch.rand(d, device=dev) + 0.5
    ch = e(bj.exp) @ bj.Shift(mu) @ bj.Scale(sg)
    chs = e(bj.exp) @ bj.Shift(0.1) @ bj.Scale(0.5)
    for name, b in (("vector params", ch), ("scalar params", chs)):
        ms = kernel_ms(bj, lambda: bj.with_logabsdet_jacobian(b, x), steps=10, device=dev)
        ms2 = kernel_ms(bj, lambda: bj.with_logabsdet_jacobian(b, x, per_sample=True), steps=10, device=dev)
        print(f"d={d:5d} N={N:8d} {name}: sum {ms:.4f} ms {N*d*8/ms/1e6/8000*100:5.1f} %   per-sample {ms2:.4f} ms {N*(d*8+4)/ms2/1e6/8000*100:5.1f} %")
    st = bj.Stacked([ch], [(1, d)])
    ms3 = kernel_ms(bj, lambda: bj.with_logabsdet_jacobian(st, x, per_sample=True), steps=10, device=dev)
    print(f"        Stacked([chain]) per-sample {ms3:.4f} ms {N*(d*8+4)/ms3/1e6/8000*100:5.1f} %")
