import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import bijectors_amd as bj
from _timing import kernel_ms
dev = torch.device("cuda", 0)
e = bj.elementwise
def wall(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for d, N in ((1001, 1024), (1001, 2048), (1001, 4096), (1001, 8192), (1001, 65536), (999, 4096), (1000, 4096), (333, 12288), (101, 40000)):
    x = torch.randn(N, d, device=dev).T
    mu = torch.randn(d, device=dev); sg = torch.rand(d, device=dev) + 0.5
    ch = e(bj.exp) @ bj.Shift(mu) @ bj.Scale(sg)
    chs = e(bj.exp) @ bj.Shift(0.1) @ bj.Scale(0.5)
    f = lambda: bj.with_logabsdet_jacobian(ch, x)
    fs = lambda: bj.with_logabsdet_jacobian(chs, x)
    print(f"d={d} N={N}: vector params wall {wall(f):7.1f} us kernel {kernel_ms(bj, f, steps=10, device=dev)*1e3:7.1f} us | scalar params wall {wall(fs):7.1f} us")
