"""Every bijector family at tall columns (dim = the number of parameters of a model): errors and throughput cliffs.  2^16 columns."""
import math, os, sys, traceback
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import bijectors_amd as bj
from _timing import kernel_ms
dev = torch.device("cuda", 0)
N = 1 << 16
e = bj.elementwise
dims = [int(v) for v in sys.argv[1:]] or [333, 1001, 2048, 5000, 10001]
print("| case | " + " | ".join(str(d) for d in dims) + " |")
print("|---|" + "---|" * len(dims))
def cases(d):
    x = torch.randn(N, d, device=dev).T
    g = torch.randn(N, d, device=dev).T
    lb = torch.randn(N, device=dev)
    xu = torch.rand(N, d, device=dev).T * 0.9 + 0.05
    mu = torch.randn(d, device=dev); sg = torch.rand(d, device=dev) + 0.5
    ch = e(bj.exp) @ bj.Shift(mu) @ bj.Scale(sg)
    a_, b_ = d // 3, 2 * (d // 3)
    st = bj.Stacked([e(bj.exp) @ bj.Scale(sg[:a_]), bj.Logit(0.0, 1.0), bj.identity], [(1, a_), (a_ + 1, b_), (b_ + 1, d)])
    xs = x.clone(); xs[a_:b_] = xu[a_:b_]
    y_ch = bj.transform(ch, x)
    sx = torch.softmax(x.T, dim=1).T.contiguous().T if False else None
    out = [
        ("chain fwd (per-sample)", lambda: bj.with_logabsdet_jacobian(ch, x, per_sample=True), 2 * d * 4),
        ("chain inverse", lambda: bj.with_logabsdet_jacobian(bj.inverse(ch), y_ch, per_sample=True), 2 * d * 4),
        ("chain vjp", lambda: bj.vjp(ch, x, g, lb), 3 * d * 4),
        ("chain vjp_params", lambda: bj.vjp_params(ch, x, g, lb), 3 * d * 4),
        ("Stacked fwd", lambda: bj.with_logabsdet_jacobian(st, xs, per_sample=True), 2 * d * 4),
        ("Stacked vjp", lambda: bj.vjp(st, xs, g, lb), 3 * d * 4),
        ("logpdf(MvNormal(mu,sigma), chain)", lambda: bj.logpdf(bj.transformed(bj.MvNormal(mu, sg), ch), y_ch), d * 4),
        ("Ordered fwd", lambda: bj.with_logabsdet_jacobian(bj.OrderedBijector(), x, per_sample=True), 2 * d * 4),
        ("Ordered vjp", lambda: bj.vjp(bj.OrderedBijector(), x, g, lb), 3 * d * 4),
        ("Simplex fwd", lambda: bj.with_logabsdet_jacobian(bj.SimplexBijector(), torch.softmax(x.T, dim=1).T, per_sample=True), 2 * d * 4),
        ("BatchNorm eval", lambda: bj.with_logabsdet_jacobian(bj.InvertibleBatchNorm(mu, 0.1 * mu, mu, sg), x, per_sample=True), 2 * d * 4),
        ("Radial", lambda: bj.with_logabsdet_jacobian(bj.RadialLayer(torch.tensor([0.5], device=dev), torch.tensor([0.3], device=dev), mu), x), 2 * d * 4),
        ("Planar x2 (composed)", lambda: bj.with_logabsdet_jacobian(bj.PlanarLayer(mu / math.sqrt(d), sg / math.sqrt(d), torch.randn(1, device=dev)) @ bj.PlanarLayer(sg / math.sqrt(d), mu / math.sqrt(d), torch.randn(1, device=dev)), x), 2 * d * 4),
    ]
    return out
table = {}
for d in dims:
    try:
        cs = cases(d)
    except Exception as ex:
        print("setup failed at", d, repr(ex)[:200]); continue
    for name, fn, bps in cs:
        try:
            ms = kernel_ms(bj, fn, steps=4, device=dev)
            table.setdefault(name, {})[d] = f"{N * bps / ms / 1e6 / 8000 * 100:.0f} %"
        except Exception as ex:
            table.setdefault(name, {})[d] = "ERR " + type(ex).__name__ + ": " + str(ex)[:60]
        torch.cuda.synchronize()
for name, row in table.items():
    print("| " + name + " | " + " | ".join(row.get(d, "-") for d in dims) + " |")
