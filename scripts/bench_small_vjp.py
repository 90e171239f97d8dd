"""Pullbacks at small per-sample sizes (what HMC / ADVI differentiate in low-dimensional models).  Markdown table."""
import os, sys, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ctypes as C
import torch
import bijectors_amd as bj

dev = torch.device("cuda", 0)
lib = bj._lib.load()
ctx = bj.context(dev)


def timed(fn, reps=10):
    from _timing import kernel_ms          # clock-settling pre-roll, then the per-launch event pairs
    return kernel_ms(bj, fn, steps=reps, device=dev)


print("| pullback | size | kernel ms (2^22 columns) | alg. B/sample | GB/s | % of 8 TB/s |")
print("|---|---|---|---|---|---|")
N = 1 << 22
e = bj.elementwise
for d in tuple(int(v) for v in os.environ.get("BJX_BENCH_DIMS", "3,4,8,10").split(",")):
    x = torch.randn(N, d, device=dev).T
    g = torch.randn(N, d, device=dev).T
    lb = torch.randn(N, device=dev)
    cases = []
    xs = torch.softmax(torch.randn(N, d, device=dev), dim=1).T
    cases.append(("vjp(SimplexBijector)", bj.SimplexBijector(), xs, torch.randn(N, d - 1, device=dev).T, 2 * d - 1 + d))
    cases.append(("vjp(inverse(SimplexBijector))", bj.inverse(bj.SimplexBijector()), torch.randn(N, d - 1, device=dev).T, g, 2 * d - 1 + d - 1))
    cases.append(("vjp(OrderedBijector)", bj.OrderedBijector(), x, g, 3 * d))
    cases.append(("vjp(exp∘Shift∘Scale)", e(bj.exp) @ bj.Shift(0.1) @ bj.Scale(0.5), x, g, 3 * d))
    W8 = torch.randn(d, 8, device=dev) / math.sqrt(d)
    U8 = torch.randn(d, 8, device=dev) / math.sqrt(d)
    cases.append(("vjp(8×PlanarLayer)", bj.PlanarLayer(W8, U8, torch.randn(8, device=dev)), x, g, 3 * d))
    cases.append(("vjp(RadialLayer)", bj.RadialLayer(torch.tensor([0.5], device=dev), torch.tensor([0.3], device=dev), torch.randn(d, device=dev)), x, g, 3 * d))
    pl8 = bj.PlanarLayer(W8, U8, torch.randn(8, device=dev))
    cases.append(("vjp_params(8×PlanarLayer)", pl8, x, g, 5 * d))
    for name, b, xin, gin, words in cases:
        try:
            ms = timed((lambda: bj.vjp_params(b, xin, gin, lb)) if name.startswith("vjp_params") else (lambda: bj.vjp(b, xin, gin, lb)))
        except Exception as ex:
            print(f"| {name} | {d} | error {ex!r} | | | |")
            continue
        bps = words * 4 + 4
        gb = bps * N / (ms * 1e-3) / 1e9
        print(f"| {name} | {d} | {ms:.4f} | {bps} | {gb:.0f} | {gb / 80:.1f} |")
