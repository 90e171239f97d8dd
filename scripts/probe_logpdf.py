import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import bijectors_amd as bj
from _timing import kernel_ms
dev = torch.device("cuda", 0)
d = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = (1 << 28) // (4 * d) // 64 * 64
e = bj.elementwise
y = torch.rand(N, d, device=dev).T + 0.1
mu = torch.randn(d, device=dev); sg = torch.rand(d, device=dev) + 0.5
b = e(bj.exp) @ bj.Shift(0.1) @ bj.Scale(0.5)
cases = {
 "logpdf std base, chain": lambda: bj.logpdf(bj.transformed(bj.MvNormal(d), b), y),
 "logpdf (mu,sigma) base, chain": lambda: bj.logpdf(bj.transformed(bj.MvNormal(mu, sg), b), y),
 "logpdf std base, identity": lambda: bj.logpdf(bj.transformed(bj.MvNormal(d)), y),
 "logpdf (mu,sigma) base, identity": lambda: bj.logpdf(bj.transformed(bj.MvNormal(mu, sg)), y),
 "logabsdetjac(inverse chain) per-sample": lambda: bj.logabsdetjac(bj.inverse(b), y),
 "wlj(inverse chain) per-sample (stores x)": lambda: bj.with_logabsdet_jacobian(bj.inverse(b), y, per_sample=True),
}
for k, f in cases.items():
    ms = kernel_ms(bj, f, steps=10, device=dev)
    print(f"{k:45s} {ms:.4f} ms  {N*(d*4+4)/ms/1e6/8000*100:.1f} % (read-only bytes)")
