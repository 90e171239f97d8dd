import os, sys, cProfile, pstats, io
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bijectors_amd as bj
dev = torch.device("cuda", 0)
d, N = 5000, 16
e = bj.elementwise
x = torch.randn(N, d, device=dev).T; g = torch.randn(N, d, device=dev).T; lb = torch.randn(N, device=dev)
mu = torch.randn(d, device=dev); sg = torch.rand(d, device=dev) + 0.5
ch = e(bj.exp) @ bj.Shift(mu) @ bj.Scale(sg)
for _ in range(20): bj.vjp(ch, x, g, lb)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(200): bj.vjp(ch, x, g, lb)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(14); print(s.getvalue()[:3500])
