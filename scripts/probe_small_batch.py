"""Wall time per call (microseconds, stream-synchronised, 200 calls) at SMALL batches of tall columns: what an HMC / ADVI step does."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bijectors_amd as bj
dev = torch.device("cuda", 0)
e = bj.elementwise
def wall(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for d in (101, 1001, 5000):
    for N in (16, 256, 4096):
        x = torch.randn(N, d, device=dev).T; g = torch.randn(N, d, device=dev).T; lb = torch.randn(N, device=dev)
        mu = torch.randn(d, device=dev); sg = torch.rand(d, device=dev) + 0.5
        ch = e(bj.exp) @ bj.Shift(mu) @ bj.Scale(sg)
        a_, b_ = d // 3, 2 * (d // 3)
        st = bj.Stacked([e(bj.exp) @ bj.Scale(sg[:a_]), bj.Logit(0.0, 1.0), bj.identity], [(1, a_), (a_ + 1, b_), (b_ + 1, d)])
        xs = x.clone(); xs[a_:b_] = torch.rand(b_ - a_, N, device=dev) * 0.9 + 0.05
        print(f"d={d:5d} N={N:5d}  chain fwd {wall(lambda: bj.with_logabsdet_jacobian(ch, x)):7.1f}  chain vjp {wall(lambda: bj.vjp(ch, x, g, lb)):7.1f}  "
              f"vjp_params {wall(lambda: bj.vjp_params(ch, x, g, lb)):7.1f}  Stacked fwd {wall(lambda: bj.with_logabsdet_jacobian(st, xs)):7.1f}  Stacked vjp {wall(lambda: bj.vjp(st, xs, g, lb)):7.1f} us")
