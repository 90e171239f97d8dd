"""vjp_params of the mean-field family y = exp(mu + sigma * z) (ADVI) at the heights given on the command line, 2^20 columns."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import bijectors_amd as bj
from _timing import kernel_ms
dev = torch.device("cuda", 0)
N = 1 << 20
for d in [int(v) for v in sys.argv[1:]]:
    z = torch.randn(N, d, device=dev).T
    g = torch.randn(N, d, device=dev).T
    lb = torch.randn(N, device=dev)
    mu = torch.randn(d, device=dev); sg = torch.rand(d, device=dev) + 0.5
    b = bj.elementwise(bj.exp) @ bj.Shift(mu) @ bj.Scale(sg)
    ms = kernel_ms(bj, lambda: bj.vjp_params(b, z, g, lb), steps=10, device=dev)
    ms2 = kernel_ms(bj, lambda: bj.vjp(b, z, g, lb), steps=10, device=dev)
    byts = N * (3 * d * 4 + 4)
    print(f"d={d:5d}  vjp_params {ms:.4f} ms {byts/ms/1e6/8000*100:5.1f} %   vjp {ms2:.4f} ms {byts/ms2/1e6/8000*100:5.1f} %")
    ms3 = kernel_ms(bj, lambda: bj.row_moments(g, z), steps=10, device=dev)
    print(f"         row_moments(a, b) alone {ms3:.4f} ms {N*(2*d*4)/ms3/1e6/8000*100:5.1f} %")
